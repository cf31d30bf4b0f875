// one backward instantiation unit: dtype=f16 head_dim=256 (head dims 136..256: three single-gradient launches, one wave per SIMD)
#define TFA_T _Float16
#define TFA_D 256
#include "tfa_bwd_inst.inc"
