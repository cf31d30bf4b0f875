// one instantiation unit: dtype=f16 head_dim=128 causal=1
#define TFA_T _Float16
#define TFA_D 128
#define TFA_CAUSAL true
#include "tfa_fwd_inst.inc"
