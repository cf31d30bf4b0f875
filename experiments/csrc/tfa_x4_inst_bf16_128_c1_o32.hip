// one instantiation unit of the x4 kernel: dtype=bf16 head_dim=128 causal=1 fp32 output
#define TFA_T __bf16
#define TFA_D 128
#define TFA_CAUSAL true
#define TFA_F32OUT true
#include "tfa_x4_inst.inc"
