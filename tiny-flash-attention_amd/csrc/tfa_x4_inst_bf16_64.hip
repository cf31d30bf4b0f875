// one instantiation unit of the x4 kernel: dtype=bf16 head_dim=64
#define TFA_T __bf16
#define TFA_D 64
#include "tfa_x4_inst.inc"
