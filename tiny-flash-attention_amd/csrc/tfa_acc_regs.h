// tfa_acc_regs.h — 128 accumulator registers as hand-owned AccVGPRs a[0:127]: the gradient accumulators of the 256-wide backward kernels and the O accumulators of the 256-wide LDS-DMA forward kernel (one wave per SIMD).
//
// At head dims above 128 a wave holds 128 accumulator registers (32 resident rows x 256 columns, fp32) plus up to 128 registers of
// resident operand fragments: both belong in AccVGPRs, where MFMAs read and write them directly, and the 256 architectural VGPRs are
// left to everything VALU touches.  Left to hipcc (builtin MFMAs, or asm with "+a" operands) the allocator splits the accumulators'
// live range around the tile loop and moves all 128 registers between the two files every iteration (v_accvgpr_write/read: 280 of
// the 575 instructions of the dQ loop) — the same failure the forward's O accumulators had (tfa_fwd_il_regs.h).  So every access
// names the physical registers, and — the x4 forward kernel's rule (tfa_fwd_kernel_x4.h) — every asm statement that touches them
// lists ALL of a0..a127 as clobbered: the allocator then never parks a value of its own there, and its "a"-constrained values (the
// resident fragments) go to a128..a255.
// Hazards (nothing inside an asm string is padded by hipcc): a VALU-written A/B operand needs 2 wait states before the MFMA reads
// it (s_nop 1 inside the string); an MFMA result needs 12 wait states (8-pass) before anything but the next accumulating MFMA
// touches it (s_nop 12 in front of the read-out).  tools/audit_mfma_hazard.py checks the generated assembly at build time.
#pragma once
#include "tfa_fwd_kernel.h"

namespace tfa {

#define TFA_G_CLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define TFA_G_LIST2 "0,2,4,6,8,10,12,14,16,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64,66,68,70,72,74,76,78,80,82,84,86,88,90,92,94,96,98,100,102,104,106,108,110,112,114,116,118,120,122,124,126"
#define TFA_G_LIST "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127"

template <typename T> struct GMfma;
template <> struct GMfma<__bf16> { static constexpr bool bf = true; };
template <> struct GMfma<_Float16> { static constexpr bool bf = false; };

#define TFA_G_ASM(PRE, OP, LO, HI) asm volatile(PRE OP " a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(a), "v"(b) : TFA_G_CLOB)
#define TFA_G_CASE(DI, LO, HI)                                                                 \
  if constexpr (DI == LO / 16) {                                                               \
    if constexpr (GMfma<T>::bf) {                                                              \
      if constexpr (PAD) TFA_G_ASM("s_nop 1\n\t", "v_mfma_f32_32x32x16_bf16", LO, HI);         \
      else TFA_G_ASM("", "v_mfma_f32_32x32x16_bf16", LO, HI);                                  \
    } else {                                                                                   \
      if constexpr (PAD) TFA_G_ASM("s_nop 1\n\t", "v_mfma_f32_32x32x16_f16", LO, HI);          \
      else TFA_G_ASM("", "v_mfma_f32_32x32x16_f16", LO, HI);                                   \
    }                                                                                          \
  }
// grad[column tile DI] += A . B.  PAD: the two wait states a VALU-written operand needs in front of the MFMA (the first MFMA behind
// the code that packed B; the others read operands that LDS reads or earlier VALU code delivered — the build's audit checks it)
template <typename T, int DI, bool PAD, typename X8> static __device__ __forceinline__ void g_mfma(X8 a, X8 b) {
  TFA_G_CASE(DI, 0, 15)
  TFA_G_CASE(DI, 16, 31)
  TFA_G_CASE(DI, 32, 47)
  TFA_G_CASE(DI, 48, 63)
  TFA_G_CASE(DI, 64, 79)
  TFA_G_CASE(DI, 80, 95)
  TFA_G_CASE(DI, 96, 111)
  TFA_G_CASE(DI, 112, 127)
}
template <typename T, bool PAD, typename X8> static __device__ __forceinline__ void g_mfma_d(int d, X8 a, X8 b) {   // d folds to a constant (unrolled loops)
  if (d == 0) g_mfma<T, 0, PAD>(a, b);
  else if (d == 1) g_mfma<T, 1, PAD>(a, b);
  else if (d == 2) g_mfma<T, 2, PAD>(a, b);
  else if (d == 3) g_mfma<T, 3, PAD>(a, b);
  else if (d == 4) g_mfma<T, 4, PAD>(a, b);
  else if (d == 5) g_mfma<T, 5, PAD>(a, b);
  else if (d == 6) g_mfma<T, 6, PAD>(a, b);
  else g_mfma<T, 7, PAD>(a, b);
}
static __device__ __forceinline__ void g_zero() {
  asm volatile(".irp r," TFA_G_LIST "\n\tv_accvgpr_write_b32 a[\\r], 0\n\t.endr" ::: TFA_G_CLOB);
}
// every accumulator *= alpha (per lane): the forward's rare "row maximum moved" path.  In front: the pending MFMAs' 12 wait states;
// behind: the two a VALU-written AccVGPR needs before an MFMA reads it.
static __device__ __forceinline__ void g_scale(float alpha) {
  float t0, t1;
  asm volatile("s_nop 12\n\t.irp r," TFA_G_LIST2 "\n\tv_accvgpr_read_b32 %0, a[\\r]\n\tv_accvgpr_read_b32 %1, a[\\r+1]\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
               "v_accvgpr_write_b32 a[\\r], %0\n\tv_accvgpr_write_b32 a[\\r+1], %1\n\t.endr\n\ts_nop 1"
               : "=&v"(t0), "=&v"(t1) : "v"(alpha) : TFA_G_CLOB);
}
// out[r] = grad[column tile DI][r]   (the s_nop covers the MFMA-write -> read distance; cold path)
#define TFA_GR(B, K) "v_accvgpr_read_b32 %" #K ", a[" #B "+" #K "]\n\t"
#define TFA_GREAD_CASE(DI, B)                                                                                                             \
  if constexpr (DI == B / 16)                                                                                                             \
    asm volatile("s_nop 12\n\t" TFA_GR(B, 0) TFA_GR(B, 1) TFA_GR(B, 2) TFA_GR(B, 3) TFA_GR(B, 4) TFA_GR(B, 5) TFA_GR(B, 6) TFA_GR(B, 7)    \
                 TFA_GR(B, 8) TFA_GR(B, 9) TFA_GR(B, 10) TFA_GR(B, 11) TFA_GR(B, 12) TFA_GR(B, 13) TFA_GR(B, 14) TFA_GR(B, 15)             \
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),                 \
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])            \
                 :: TFA_G_CLOB);
template <int DI> static __device__ __forceinline__ void g_read(float (&o)[16]) {
  TFA_GREAD_CASE(DI, 0)
  TFA_GREAD_CASE(DI, 16)
  TFA_GREAD_CASE(DI, 32)
  TFA_GREAD_CASE(DI, 48)
  TFA_GREAD_CASE(DI, 64)
  TFA_GREAD_CASE(DI, 80)
  TFA_GREAD_CASE(DI, 96)
  TFA_GREAD_CASE(DI, 112)
}
static __device__ __forceinline__ void g_read_d(int d, float (&o)[16]) {
  if (d == 0) g_read<0>(o);
  else if (d == 1) g_read<1>(o);
  else if (d == 2) g_read<2>(o);
  else if (d == 3) g_read<3>(o);
  else if (d == 4) g_read<4>(o);
  else if (d == 5) g_read<5>(o);
  else if (d == 6) g_read<6>(o);
  else g_read<7>(o);
}

}  // namespace tfa
