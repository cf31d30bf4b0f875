// one instantiation unit: dtype=bf16 head_dim=64 causal=0
#define TFA_T __bf16
#define TFA_D 64
#define TFA_CAUSAL false
#include "tfa_fwd_inst.inc"
