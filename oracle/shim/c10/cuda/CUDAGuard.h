/* empty stand-in so the reference CPU path (flash_attention_c/csrc/attn.h:1-4) compiles without a CUDA toolkit;
   none of these headers' symbols are used by attn.cpp. Test infrastructure only. */
