// one instantiation unit of the x4 kernel: dtype=bf16 head_dim=64 causal=0 fp32 output
#define TFA_T __bf16
#define TFA_D 64
#define TFA_CAUSAL false
#define TFA_F32OUT true
#include "tfa_x4_inst.inc"
