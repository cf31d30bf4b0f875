// one backward instantiation unit: dtype=bf16, 256-wide kernels with 5 valid 32-column blocks (head dims 136..160)
#define TFA_T __bf16
#define TFA_D 256
#define TFA_DVB 5
#include "tfa_bwd_inst.inc"
