// one instantiation unit of the x4 kernel: dtype=f16 head_dim=128 causal=0 fp32 output
#define TFA_T _Float16
#define TFA_D 128
#define TFA_CAUSAL false
#define TFA_F32OUT true
#include "tfa_x4_inst.inc"
