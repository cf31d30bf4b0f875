// one instantiation unit of the x4 kernel: dtype=bf16 head_dim=128
#define TFA_T __bf16
#define TFA_D 128
#include "tfa_x4_inst.inc"
