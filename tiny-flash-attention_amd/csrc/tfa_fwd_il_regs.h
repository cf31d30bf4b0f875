// tfa_fwd_il_regs.h — the O accumulators of the issue-interleaved forward kernels as hand-pinned registers (gfx950).
// Part of tfa_fwd_kernel_il.h (split out in round 3; no code change).
#pragma once
#include "tfa_fwd_kernel_dma.h"

namespace tfa {

// ---- O accumulators in hand-pinned registers v[192:255] ---------------------------------------------------------
// The kernel is compiled with amdgpu_num_vgpr(96) (LLVM doubles the request on gfx90a+: 192 unified registers): the register allocator owns v0..v191 and never sees O.  With O as
// ordinary SSA values (builtin MFMA) the allocator split the 64-register live range around the loop and copied all of O
// between two register sets every iteration; with "+a" (AGPR) operands it halves the VGPR budget to 128.  Every access
// to O is therefore inline asm naming the physical registers: d tile i lives in v[192+16i : 207+16i].
#define TFA_O_CLOB0 "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207"
#define TFA_O_CLOB1 "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223"
#define TFA_O_CLOB2 "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239"
#define TFA_O_CLOB3 "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define TFA_O_LIST01 "192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223"
#define TFA_O_LIST23 "224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255"
template <typename T> struct MfmaName;
template <> struct MfmaName<__bf16> { static constexpr bool bf = true; };
template <> struct MfmaName<_Float16> { static constexpr bool bf = false; };

// O[d tile DI] += A.B.  A VALU write of an A/B operand needs 2 wait states before an MFMA reads it and the compiler
// cannot see that this asm is an MFMA: every caller packs P at least one whole MFMA slot before the MFMA that reads it.
#define TFA_PV_CASE(DI, LO, HI, CLOB)                                                                                  \
  if constexpr (DI == (LO - 192) / 16) {                                                                               \
    if constexpr (MfmaName<T>::bf)                                                                                     \
      asm volatile("v_mfma_f32_32x32x16_bf16 v[" #LO ":" #HI "], %0, %1, v[" #LO ":" #HI "]" ::"v"(a), "v"(b) : CLOB); \
    else                                                                                                               \
      asm volatile("v_mfma_f32_32x32x16_f16 v[" #LO ":" #HI "], %0, %1, v[" #LO ":" #HI "]" ::"v"(a), "v"(b) : CLOB);  \
  }
template <typename T, int DI, typename X8> static __device__ __forceinline__ void o_mfma(X8 a, X8 b) {
  TFA_PV_CASE(DI, 192, 207, TFA_O_CLOB0)
  TFA_PV_CASE(DI, 208, 223, TFA_O_CLOB1)
  TFA_PV_CASE(DI, 224, 239, TFA_O_CLOB2)
  TFA_PV_CASE(DI, 240, 255, TFA_O_CLOB3)
}
template <typename T, typename X8> static __device__ __forceinline__ void o_mfma_d(int d, X8 a, X8 b) {   // d folds to a constant
  if (d == 0) o_mfma<T, 0>(a, b);
  else if (d == 1) o_mfma<T, 1>(a, b);
  else if (d == 2) o_mfma<T, 2>(a, b);
  else o_mfma<T, 3>(a, b);
}
template <int DT> static __device__ __forceinline__ void o_zero() {
  asm volatile(".irp r," TFA_O_LIST01 "\n\tv_mov_b32 v[\\r], 0\n\t.endr" ::: TFA_O_CLOB0, TFA_O_CLOB1);
  if constexpr (DT > 2) asm volatile(".irp r," TFA_O_LIST23 "\n\tv_mov_b32 v[\\r], 0\n\t.endr" ::: TFA_O_CLOB2, TFA_O_CLOB3);
}
// O *= alpha (per lane); the leading s_nops cover the MFMA-write -> VALU-read distance (cold path)
template <int DT> static __device__ __forceinline__ void o_scale(float alpha) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t.irp r," TFA_O_LIST01 "\n\tv_mul_f32 v[\\r], v[\\r], %0\n\t.endr" ::"v"(alpha) : TFA_O_CLOB0, TFA_O_CLOB1);
  if constexpr (DT > 2)
    asm volatile(".irp r," TFA_O_LIST23 "\n\tv_mul_f32 v[\\r], v[\\r], %0\n\t.endr" ::"v"(alpha) : TFA_O_CLOB2, TFA_O_CLOB3);
}
// (two v_pk_mul_f32 per quad instead: -5 %, profiles/r05_exact_il8.txt — packed fp32 forms do not run in an MFMA's shadow)
// O registers 192 + 4 Q .. 192 + 4 Q + 3 *= alpha: the exact-max body's share of the re-base behind QK^T MFMA Q (the PV MFMAs that wrote O are a tile
// barrier away; the next ones that read it as C are >= one MFMA slot behind the last of these: tools/audit_mfma_hazard.py checks both at build time)
template <int Q> static __device__ __forceinline__ void o_scale4(float alpha) {
  asm volatile("v_mul_f32 v[192+4*%1], v[192+4*%1], %0\n\tv_mul_f32 v[193+4*%1], v[193+4*%1], %0\n\t"
               "v_mul_f32 v[194+4*%1], v[194+4*%1], %0\n\tv_mul_f32 v[195+4*%1], v[195+4*%1], %0" ::"v"(alpha), "n"(Q)
               : TFA_O_CLOB0, TFA_O_CLOB1, TFA_O_CLOB2, TFA_O_CLOB3);
}
static __device__ __forceinline__ void o_scale4_d(int q, float alpha) {   // q folds to a constant
  switch (q) {
    case 0: o_scale4<0>(alpha); break; case 1: o_scale4<1>(alpha); break; case 2: o_scale4<2>(alpha); break; case 3: o_scale4<3>(alpha); break;
    case 4: o_scale4<4>(alpha); break; case 5: o_scale4<5>(alpha); break; case 6: o_scale4<6>(alpha); break; case 7: o_scale4<7>(alpha); break;
    case 8: o_scale4<8>(alpha); break; case 9: o_scale4<9>(alpha); break; case 10: o_scale4<10>(alpha); break; case 11: o_scale4<11>(alpha); break;
    case 12: o_scale4<12>(alpha); break; case 13: o_scale4<13>(alpha); break; case 14: o_scale4<14>(alpha); break; default: o_scale4<15>(alpha); break;
  }
}
// out[r] = O[d tile DI][r] * inv
#define TFA_OR(B, K) "v_mul_f32 %" #K ", v[" #B "+" #K "], %16\n\t"
#define TFA_OREAD_CASE(DI, B)                                                                                          \
  if constexpr (DI == (B - 192) / 16)                                                                                  \
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" TFA_OR(B, 0) TFA_OR(B, 1) TFA_OR(B, 2) TFA_OR(B, 3) TFA_OR(B, 4) TFA_OR(B, 5) TFA_OR(B, 6)       \
                 TFA_OR(B, 7) TFA_OR(B, 8) TFA_OR(B, 9) TFA_OR(B, 10) TFA_OR(B, 11) TFA_OR(B, 12) TFA_OR(B, 13) TFA_OR(B, 14) TFA_OR(B, 15) \
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),  \
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])\
                 : "v"(inv));
template <int DI> static __device__ __forceinline__ void o_read(float (&o)[16], float inv) {
  TFA_OREAD_CASE(DI, 192)
  TFA_OREAD_CASE(DI, 208)
  TFA_OREAD_CASE(DI, 224)
  TFA_OREAD_CASE(DI, 240)
}

}  // namespace tfa
