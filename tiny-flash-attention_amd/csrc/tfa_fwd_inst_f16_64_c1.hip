// one instantiation unit: dtype=f16 head_dim=64 causal=1
#define TFA_T _Float16
#define TFA_D 64
#define TFA_CAUSAL true
#include "tfa_fwd_inst.inc"
