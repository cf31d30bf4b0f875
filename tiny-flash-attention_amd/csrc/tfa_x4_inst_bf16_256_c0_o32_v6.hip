// one instantiation unit of the x4 kernel: dtype=bf16, 256 wide with 6 valid 32-column blocks (head dims 168..192), causal=0, fp32 output
#define TFA_T __bf16
#define TFA_D 256
#define TFA_CAUSAL false
#define TFA_F32OUT true
#define TFA_DVB 6
#include "tfa_x4_inst.inc"
