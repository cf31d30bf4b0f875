// one instantiation unit of the x4 kernel: dtype=f16 head_dim=256 (one 32-row block per wave)
#define TFA_T _Float16
#define TFA_D 256
#include "tfa_x4_inst.inc"
