// one instantiation unit: dtype=f16 head_dim=128 causal=0
#define TFA_T _Float16
#define TFA_D 128
#define TFA_CAUSAL false
#include "tfa_fwd_inst.inc"
