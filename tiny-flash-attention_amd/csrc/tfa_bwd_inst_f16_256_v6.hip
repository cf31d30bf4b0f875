// one backward instantiation unit: dtype=f16, 256-wide kernels with 6 valid 32-column blocks (head dims 168..192)
#define TFA_T _Float16
#define TFA_D 256
#define TFA_DVB 6
#include "tfa_bwd_inst.inc"
