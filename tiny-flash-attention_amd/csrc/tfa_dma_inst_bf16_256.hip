// the 256-wide LDS-DMA kernel (split-KV partial pass of head dims above 128), dtype=bf16
#define TFA_T __bf16
#include "tfa_dma256_inst.inc"
