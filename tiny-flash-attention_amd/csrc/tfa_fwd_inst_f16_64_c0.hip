// one instantiation unit: dtype=f16 head_dim=64 causal=0
#define TFA_T _Float16
#define TFA_D 64
#define TFA_CAUSAL false
#include "tfa_fwd_inst.inc"
