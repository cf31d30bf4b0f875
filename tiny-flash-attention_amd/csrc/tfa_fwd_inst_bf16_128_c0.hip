// one instantiation unit: dtype=bf16 head_dim=128 causal=0
#define TFA_ABLATE 1   // timing-only ablation kernels live in this unit only
#define TFA_T __bf16
#define TFA_D 128
#define TFA_CAUSAL false
#include "tfa_fwd_inst.inc"
