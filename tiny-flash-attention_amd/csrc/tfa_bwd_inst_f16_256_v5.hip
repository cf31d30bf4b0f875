// one backward instantiation unit: dtype=f16, 256-wide kernels with 5 valid 32-column blocks (head dims 136..160)
#define TFA_T _Float16
#define TFA_D 256
#define TFA_DVB 5
#include "tfa_bwd_inst.inc"
