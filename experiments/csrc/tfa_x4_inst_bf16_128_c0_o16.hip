// one instantiation unit of the x4 kernel: dtype=bf16 head_dim=128 causal=0 16-bit output
#define TFA_T __bf16
#define TFA_D 128
#define TFA_CAUSAL false
#define TFA_F32OUT false
#include "tfa_x4_inst.inc"
