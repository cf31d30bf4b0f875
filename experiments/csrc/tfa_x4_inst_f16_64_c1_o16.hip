// one instantiation unit of the x4 kernel: dtype=f16 head_dim=64 causal=1 16-bit output
#define TFA_T _Float16
#define TFA_D 64
#define TFA_CAUSAL true
#define TFA_F32OUT false
#include "tfa_x4_inst.inc"
