// one instantiation unit of the x4 kernel: dtype=f16 head_dim=64
#define TFA_T _Float16
#define TFA_D 64
#include "tfa_x4_inst.inc"
