// one instantiation unit of the x4 kernel: dtype=f16, 256 wide with 5 valid 32-column blocks (head dims 136..160), causal=0, fp32 output
#define TFA_T _Float16
#define TFA_D 256
#define TFA_CAUSAL false
#define TFA_F32OUT true
#define TFA_DVB 5
#include "tfa_x4_inst.inc"
