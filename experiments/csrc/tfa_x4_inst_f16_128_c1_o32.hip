// one instantiation unit of the x4 kernel: dtype=f16 head_dim=128 causal=1 fp32 output
#define TFA_T _Float16
#define TFA_D 128
#define TFA_CAUSAL true
#define TFA_F32OUT true
#include "tfa_x4_inst.inc"
