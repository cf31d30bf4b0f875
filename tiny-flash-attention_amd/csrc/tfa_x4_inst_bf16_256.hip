// one instantiation unit of the x4 kernel: dtype=bf16 head_dim=256 (one 32-row block per wave)
#define TFA_T __bf16
#define TFA_D 256
#include "tfa_x4_inst.inc"
