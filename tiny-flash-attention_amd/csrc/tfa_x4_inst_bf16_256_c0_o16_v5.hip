// one instantiation unit of the x4 kernel: dtype=bf16, 256 wide with 5 valid 32-column blocks (head dims 136..160), causal=0, 16-bit output
#define TFA_T __bf16
#define TFA_D 256
#define TFA_CAUSAL false
#define TFA_F32OUT false
#define TFA_DVB 5
#include "tfa_x4_inst.inc"
