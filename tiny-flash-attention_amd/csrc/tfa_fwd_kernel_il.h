// tfa_fwd_kernel_il.h — forward kernel with an ISSUE-INTERLEAVED tile loop (gfx950).
//
// Why (tools/probe_issue.hip, tools/probe_overlap.hip, profiles/r01_pmc_default_cfg3.txt):
//   * one SIMD has ONE VALU issue port; a v_mfma_f32_32x32x16 keeps the matrix pipe busy for 32 cycles but
//     the port only for ~4-8, so ~5 ordinary VALU instructions (a v_exp_f32 counts as 2) issued by the SAME
//     wave right behind an MFMA run for free in its shadow;
//   * a wave that streams MFMAs back to back re-arms the port the moment it frees, so VALU work of the
//     OTHER wave on that SIMD is starved (head-of-line), and s_setprio does not change that;
//   * in the burst-structured kernels (QK^T burst, softmax burst, PV burst) the PMC counters show
//     time ~= MFMA time + VALU time: nothing overlaps.
// So the loop below is software-pipelined inside each wave and the order is pinned MFMA by MFMA with
// sched_barrier(0): every MFMA is followed by its share of the softmax VALU work of ANOTHER tile.
//
//   iteration j:   part 1   S(j+1) = K(j+1) Q^T   (2*D/16 MFMAs)  each followed by exp2/sum/pack of ~1.25 elements of tile j
//                  part 2   O += P(j) V(j)        (4*D/32 MFMAs)  each followed by 1 element of tile j (first 3/4) and
//                                                                  one v_max3 of the row max of S(j+1)
//   P(j) slot s (16 keys) is complete before the first PV MFMA that consumes it.  K tiles run one tile
//   ahead of V tiles in LDS (iteration j reads K(j+1) and V(j)); two K and two V buffers, refilled by
//   LDS-DMA at the top of the iteration that follows their last read; one workgroup barrier per tile.
//   The loop is unrolled by two so buffer addresses are immediates and S(j)/S(j+1) swap without copies.
//   Tiles that need masking (causal diagonal, ragged tail) run the same body with the mask applied
//   between the two parts (template flag, real branch: never if-converted into the steady state).
//
// Three forms are dispatched (tfa_api.hip:pick_variant): 8 waves x 32 rows = 256-row blocks, one workgroup per CU (il8, the
// default); 4 waves = 128-row blocks, two workgroups per CU (il4, medium grids); and the KEY-SPLIT form for grids of at most
// one 128-row block per CU (VF_IL_KSPLIT: 8 waves on one 128-row block, the two groups of four waves take the even / odd
// KV tiles and merge through LDS).  il8 and il4 also exist with per-tile buffer descriptors for head slices of 2 GiB and
// more (VF_IL_WINDOWED).
#pragma once
#include <type_traits>
#include "tfa_fwd_kernel_dma.h"

namespace tfa {

constexpr int VF_IL = 32768;        // issue-interleaved kernel (this file)
constexpr int VF_IL_DMASPREAD = 65536;   // issue the LDS-DMA pieces between MFMAs of part 1 instead of at the top
constexpr int VF_IL_EPI = 262144;        // 16-bit O leaves through a separate LDS region as whole rows (16-byte coalesced stores)
constexpr int VF_IL_EPI_INPLACE = 1048576;   // with EPI: the epilogue slices live in the (idle) tile buffers instead of a separate
                                             // region — 64 KiB total, so two 4-wave workgroups still fit a CU
constexpr int VF_IL_PREF = 524288;       // the next pass's K(0)/V(0)/K(1)/Q are requested BEFORE this pass's epilogue
constexpr int VF_IL_SEAM = 2097152;      // causal pairs: the heavy pass's last two iterations already request the light pass's K(0), K(1), V(0)
                                         // (same head, same K/V: the tile stream simply continues across the seam) and its Q
                                         // fragments are requested before the epilogue: the second prologue finds everything on chip
constexpr int VF_IL_KSPLIT = 1 << 25;    // small non-causal grids: the 8 waves work on ONE 128-row query block — waves 0-3 ("group 0") take the even
                                         // KV tiles, waves 4-7 the odd ones, each group with its own K/V ring in LDS; group 1 hands its O, m, l
                                         // to group 0 through LDS at the end.  Two waves per SIMD where 128-row workgroups alone would leave one.
constexpr int VF_IL_IDLE = 1 << 26;      // waves whose 32 rows all lie behind the last query row skip the tile work (decode-like problems: one
                                         // query block with one valid wave).  Its own instantiation: the flag costs the causal headline 0.5 %
#ifndef TFA_IL_DECODE_NT
#define TFA_IL_DECODE_NT 1               // decode instantiations (VF_IL_IDLE: one query block per head): K/V tiles nobody else reads are
                                         // loaded non-temporal — MHA decode 6.3 -> 6.6 TB/s, packed GQA 5.9 -> 7.0 TB/s (profiles/r02_decode_nt_ab.txt)
#endif
constexpr int VF_IL_WINDOWED = 1 << 24;  // K/V tiles through per-tile descriptors (rsrc_at): a (b,h) slice may exceed 2 GiB.  ~6 % slower (a fresh
                                         // descriptor per tile: ~14 SALU + the SGPR->VMEM wait states), so only launched when needed
constexpr int VF_IL_DMASTAGGER = 131072; // with DMASPREAD: the upper half of the waves issues its pieces behind the first PV MFMAs,
                                         // so the two waves of a SIMD never sit in an LDS-DMA issue stall at the same time

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
// S accumulators of the fast path: the MFMA is inline asm only so that it stays the FIRST instruction of its slot (a
// builtin MFMA may be scheduled behind the slot's VALU work, which then delays the matrix pipe instead of hiding under
// it).  The results are first read by VALU code at least two MFMA issues later (row max in part 2), which covers
// the MFMA-write -> VALU-read distance the compiler cannot insert for an asm.
template <typename T, typename X8> static __device__ __forceinline__ void s_mfma0(f32x16& c, X8 a, X8 b) {
  if constexpr (MfmaName<T>::bf) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
template <typename T, typename X8> static __device__ __forceinline__ void s_mfma(f32x16& c, X8 a, X8 b) {
  if constexpr (MfmaName<T>::bf) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <typename T, typename X8> static __device__ __forceinline__ void o_mfma_d(int d, X8 a, X8 b) {   // d folds to a constant
  if (d == 0) o_mfma<T, 0>(a, b);
  else if (d == 1) o_mfma<T, 1>(a, b);
  else if (d == 2) o_mfma<T, 2>(a, b);
  else o_mfma<T, 3>(a, b);
}
template <int DT> static __device__ __forceinline__ void o_zero() {
  asm volatile(".irp r," TFA_O_LIST01 "\n\tv_mov_b32 v[\\r], 0\n\t.endr" ::: TFA_O_CLOB0, TFA_O_CLOB1);
  if constexpr (DT == 4) asm volatile(".irp r," TFA_O_LIST23 "\n\tv_mov_b32 v[\\r], 0\n\t.endr" ::: TFA_O_CLOB2, TFA_O_CLOB3);
}
// O *= alpha (per lane); the leading s_nops cover the MFMA-write -> VALU-read distance (cold path)
template <int DT> static __device__ __forceinline__ void o_scale(float alpha) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t.irp r," TFA_O_LIST01 "\n\tv_mul_f32 v[\\r], v[\\r], %0\n\t.endr" ::"v"(alpha) : TFA_O_CLOB0, TFA_O_CLOB1);
  if constexpr (DT == 4)
    asm volatile(".irp r," TFA_O_LIST23 "\n\tv_mul_f32 v[\\r], v[\\r], %0\n\t.endr" ::"v"(alpha) : TFA_O_CLOB2, TFA_O_CLOB3);
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

// AB: timing-only ablation bits of the fast path (results are wrong when set; tools/ablate_il.py)
constexpr int ILAB_NOEXP = 1, ILAB_NODMA = 2, ILAB_NOBARRIER = 4, ILAB_NOMAX = 8, ILAB_NOQK = 16, ILAB_NOPV = 32, ILAB_NOKREAD = 64, ILAB_NOVREAD = 128;

template <typename T, int D, int NW, bool CAUSAL, bool F32OUT, int VF, int AB = 0>
__global__ __launch_bounds__(NW * 64, 2) __attribute__((amdgpu_num_vgpr(96))) void fwd_kernel_il(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr bool KSPLIT = (VF & VF_IL_KSPLIT) != 0;
  constexpr int NWG = KSPLIT ? NW / 2 : NW;        // waves per query block (KSPLIT: per key-tile group)
  constexpr int BM = NWG * 32;
  static_assert(!KSPLIT || ((VF & VF_IL_EPI_INPLACE) && !(VF & (VF_IL_SEAM | VF_IL_PREF | VF_IL_WINDOWED))), "KSPLIT: in-place epilogue");
  constexpr int KSTEP = KSPLIT ? 2 : 1;            // a wave's tile t is tile KSTEP*t + grp of the head
  constexpr int BN = 64;
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NWG;                // DMA pieces per wave per tensor per tile
  constexpr int DS = D / 16;
  constexpr int DT = D / 32;
  constexpr int N1 = 2 * DS;                       // QK^T MFMAs per tile
  constexpr int N2 = 4 * DT;                       // PV MFMAs per tile
#ifndef TFA_IL_NE1
#define TFA_IL_NE1 21
#endif
#ifndef TFA_IL_PF
#define TFA_IL_PF 2
#endif
  constexpr int NE1 = TFA_IL_NE1;                  // softmax elements (of 32 per lane) summed/packed during part 1
  constexpr int PFK = TFA_IL_PF, PFV = TFA_IL_PF;  // fragment read-ahead, in MFMAs
#ifndef TFA_IL_UNIFORM
#define TFA_IL_UNIFORM 0
#endif
#ifndef TFA_IL_ASMQK
#define TFA_IL_ASMQK 0
#endif
  // MFMA slot (0..N1+N2-1) in which softmax element e (0..31) is summed and packed; its exp2 is issued one slot and its
  // scale/subtract two slots earlier.  P slot s (elements 8s..8s+7) feeds PV MFMAs N1+DT*s.., so it must be packed in
  // an EARLIER slot than N1+DT*s (also the distance the asm MFMA needs after a VALU write of its operand).
  auto slot_of_elem = [](int e) constexpr -> int {
#if TFA_IL_UNIFORM
    return 1 + e * (N1 + 3 * DT - 1) / 32;           // evenly over the slots before the last P slot is consumed
#else
    return 1 + (e < NE1 ? e * N1 / NE1 : N1 + (e - NE1) * (3 * DT - 1) / (32 - NE1));
#endif
  };
  static_assert(slot_of_elem(7) < N1 && slot_of_elem(15) < N1 + DT && slot_of_elem(23) < N1 + 2 * DT && slot_of_elem(31) < N1 + 3 * DT,
                "a P slot is packed too late for the PV MFMA that reads it");
#ifndef TFA_IL_QKSPLIT
#define TFA_IL_QKSPLIT 0
#endif
  // QK^T MFMA i works on key block KT(i) with k-slot KS(i): interleaved (0) or one key block after the other (1)
  constexpr bool QKSPLIT = TFA_IL_QKSPLIT;
#define KT(i) (QKSPLIT ? (i) / DS : (i) & 1)
#define KS(i) (QKSPLIT ? (i) % DS : (i) >> 1)
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  static_assert(PPW >= 1 && PPW * NWG == PIECES, "tile does not split into whole DMA pieces per wave");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = KSPLIT ? wave_id / NWG : 0;      // key-tile group of this wave (KSPLIT), its own four tile buffers
  char* const gsm = smem + grp * 4 * TILE_BYTES;
  char* const kl = gsm;                            // K buffers 0,1
  char* const vl = gsm + 2 * TILE_BYTES;           // V buffers 0,1
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)gsm;

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, rt_start = 0;
  if (p.trace) { rt_start = __builtin_amdgcn_s_memrealtime(); t_start = __builtin_amdgcn_s_memtime(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = KSPLIT ? wave_id % NWG : wave_id;   // index inside the query block: rows, DMA pieces, epilogue slice
#ifndef TFA_IL_WPERM
#define TFA_IL_WPERM 0
#endif
  // the 32-row block of a wave.  Causal, 8 waves: waves w and w+4 share a SIMD and the diagonal block gives row block r
  // ceil((r+1)/2) tiles — with the identity the SIMDs get 4,4,6,6 tiles, with {0,1,2,3,7,6,5,4} they get 5,4,5,4
  const int wrow = (TFA_IL_WPERM && CAUSAL && NW == 8 && wave >= 4) ? 11 - wave : wave;
  const int qi = lane & 31;
  const int hi = lane >> 5;

  int bh, wi;
  {
    const int id = blockIdx.x;
    const int G = p.H / p.Hk;
    if (G > 1 && ((p.B * p.Hk) & 7) == 0) {
      // GQA: the G query heads of a K/V head stay on ONE XCD (its K/V tiles are fetched into one L2 instead of G), K/V heads
      // round-robin over the XCDs: XCD x works through K/V heads x, x+8, ... and, inside one, through its query heads
      const int x = id & 7, s = id >> 3;
      const int per = G * p.nwork, kg = x + 8 * (s / per), r = s % per;
      bh = (kg / p.Hk) * p.H + (kg % p.Hk) * G + r / p.nwork;
      wi = r % p.nwork;
    } else if ((p.nbh & 7) == 0) {
      const int x = id & 7, s = id >> 3;
      bh = x + 8 * (s / p.nwork);
      wi = s % p.nwork;
    } else {
      bh = id / p.nwork;
      wi = id % p.nwork;
    }
  }
  const int b = bh / p.H;
  const int h = bh - b * p.H;
  const int hk = h / (p.H / p.Hk);
  const int shift = p.shift;

  const T* qbase = reinterpret_cast<const T*>(p.q) + b * p.qs_b + h * p.qs_h;
  const T* kbase = reinterpret_cast<const T*>(p.k) + b * p.ks_b + hk * p.ks_h;
  const T* vbase = reinterpret_cast<const T*>(p.v) + b * p.vs_b + hk * p.vs_h;
  // Q/O: one descriptor per query block (rsrc_at, once per pass).  K/V: one per slice, or — VF_IL_WINDOWED — one per tile.
  constexpr bool WIN = (VF & VF_IL_WINDOWED) != 0;
  // KSPLIT: group g sees the key sequence through a strided view — its tile j is tile 2j+g of the head
  unsigned long long k_bytes = p.k_bytes, v_bytes = p.v_bytes;
  if (KSPLIT) {
    const unsigned long long ko = (unsigned long long)grp * BN * p.ks_n * 2, vo = (unsigned long long)grp * BN * p.vs_n * 2;
    kbase += grp * BN * p.ks_n;
    vbase += grp * BN * p.vs_n;
    k_bytes = k_bytes > ko ? k_bytes - ko : 0;
    v_bytes = v_bytes > vo ? v_bytes - vo : 0;
  }
  auto k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, WIN ? 0u : (unsigned)k_bytes, 0x00020000);
  auto v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, WIN ? 0u : (unsigned)v_bytes, 0x00020000);

  int k_src[PPW], v_src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave * PPW + i;
    {
      const int row = pc * (1024 / (D * 2)) + lane / CPR;
      const int cpos = lane % CPR;
      const int kch = cpos ^ k_swz<D>(row);            // source chunk of this lane; chunks beyond the valid head dim read as zeros
      k_src[i] = kch * 8 < p.dv ? row * (int)p.ks_n * 2 + (kch << 4) : (int)TFA_OOB;
    }
    {
      const int o = pc * 1024 + lane * 16;
      const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
      const int dt = sub % DT, sh = sub / DT;
      const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
      v_src[i] = (dt * 4 + pcs) * 8 < p.dv ? key * (int)p.vs_n * 2 + ((dt * 4 + pcs) << 4) : (int)TFA_OOB;
    }
  }
  const int k_tile_stride = (KSPLIT ? 2 : 1) * BN * (int)p.ks_n * 2;
  const int v_tile_stride = (KSPLIT ? 2 : 1) * BN * (int)p.vs_n * 2;

  const bool kv_private = p.H == p.Hk;             // (query heads that share a K/V head share its tiles in L2: no streaming hint then)
  auto dma_k1 = [&](int t, int buf, int i) {
    if constexpr (WIN) lds_dma16_m0_fresh(rsrc_at(kbase, k_bytes, (unsigned long long)t * (unsigned)k_tile_stride), lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i]);
    else if ((VF & VF_IL_IDLE) && TFA_IL_DECODE_NT && kv_private) lds_dma16_m0_nt(k_rs, lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i] + t * k_tile_stride);
    else lds_dma16_m0(k_rs, lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i] + t * k_tile_stride);
  };
  auto dma_v1 = [&](int t, int buf, int i) {
    if constexpr (WIN) lds_dma16_m0_fresh(rsrc_at(vbase, v_bytes, (unsigned long long)t * (unsigned)v_tile_stride), lds_base + (2 + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i]);
    else if ((VF & VF_IL_IDLE) && TFA_IL_DECODE_NT && kv_private) lds_dma16_m0_nt(v_rs, lds_base + (2 + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i] + t * v_tile_stride);
    else lds_dma16_m0(v_rs, lds_base + (2 + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i] + t * v_tile_stride);
  };
  auto dma_k = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_k1(t, buf, i);
  };
  auto dma_v = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_v1(t, buf, i);
  };

  // K fragment of k-slot sl: 16-byte chunk (2*sl + hi) ^ swz of row qi.  2*sl and hi do not share bits, so the LDS
  // address is (row base + ((hi ^ swz) << 4)) ^ (sl << 5): one register and one v_xor per read pair (smem is 1 KiB aligned).
  unsigned k_rd_addr = lds_base + qi * (D * 2) + ((hi ^ k_swz<D>(qi)) << 4);
  asm volatile("" : "+v"(k_rd_addr));
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const float sc = p.scale_log2;
  int nt_total = 0, n_slow = 0;
  unsigned long long tw_wait = 0, tw_bar = 0;   // TFA_IL_TRACEWAIT debug sums

  const int npass = PAIR ? ((p.nmb - 1 - wi) != wi ? 2 : 1) : 1;
  constexpr bool EPI = (VF & VF_IL_EPI) != 0;
  constexpr bool PREF = (VF & VF_IL_PREF) != 0;
  constexpr bool SEAM = PAIR && (VF & VF_IL_SEAM) != 0;
  static_assert(!(SEAM && PREF), "SEAM includes the Q prefetch");
  static_assert(!(SEAM && (VF & VF_IL_EPI_INPLACE)), "the in-place epilogue would overwrite the streamed tiles");
  bool seam_in = false;                              // this pass's first tiles and Q were requested by the previous pass
  X8 qf[DS];
  auto own_tiles = [&](int ntg) -> int { return KSPLIT ? (ntg - grp + 1) >> 1 : ntg; };   // this wave's share of ntg tiles of the head
  auto key0_of = [&](int t) -> int { return (KSTEP * t + grp) * BN; };                        // first key of the wave's tile t
  auto block_of = [&](int pass) -> int {
    if (PAIR) return pass == 0 ? (p.nmb - 1 - wi) : wi;
    return CAUSAL ? (p.nmb - 1 - wi) : wi;
  };
  // requests for the start of query block mbx: K(0), V(0), K(1) by LDS-DMA and this lane's Q fragments
  auto issue_prologue = [&](int mbx, bool with_dma) {
    const int q0x = mbx * BM;
    int kve = p.Nk;
    if (CAUSAL) {
      const int lim = (((VF & VF_IL_IDLE) && p.row_mod > 0) ? p.row_mod : q0x + BM) + shift;
      kve = lim < kve ? lim : kve;
    }
    const int ntx = own_tiles(kve > 0 ? (kve + BN - 1) / BN : 0);
    if (with_dma && ntx > 0) dma_k(0, 0);
    if (with_dma && ntx > 0) dma_v(0, 0);
    if (with_dma && ntx > 1) dma_k(1, 1);
    auto q_rs = rsrc_at(qbase, p.q_bytes, WIN ? (unsigned long long)q0x * (unsigned long long)p.qs_n * 2ull : 0ull);
    const int qoff = ((WIN ? 0 : q0x) + wrow * 32 + qi) * (int)p.qs_n * 2 + hi * 16;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
#ifndef TFA_IL_QLOAD_AUX
#define TFA_IL_QLOAD_AUX 0
#endif
      u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(q_rs, (2 * s + hi) * 8 < p.dv ? qoff + s * 32 : (int)TFA_OOB, 0, TFA_IL_QLOAD_AUX);
      qf[s] = __builtin_bit_cast(X8, t);
    }
  };
  const int tr_pass = (p.dbg & 128) ? 1 : 0;
#pragma nounroll
  for (int pass = 0; pass < npass; ++pass) {
    const int mb = block_of(pass);
    if (p.trace && pass == 1 && tr_pass == 1) t_start = __builtin_amdgcn_s_memtime();
    const int q0 = mb * BM;
    // query POSITIONS of the wave's rows (for the causal mask): rows themselves, or — packed GQA heads, decode instantiation
    // only — row % row_mod, in which case a wave's rows span every position 0 .. row_mod-1
    const int rmod = ((VF & VF_IL_IDLE) && CAUSAL) ? p.row_mod : 0;
    int kv_end = p.Nk;
    if (CAUSAL) {
      const int lim = (rmod > 0 ? rmod : q0 + BM) + shift;
      kv_end = lim < kv_end ? lim : kv_end;
    }
    const int ntg = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;   // tiles of the head this block visits
    const int nt_own = own_tiles(ntg);
    // iterations of the WORKGROUP (barriers, DMA): KSPLIT: group 0's count — group 1 may have one tile less (nact below)
    const int nt = KSPLIT ? (ntg + 1) >> 1 : ntg;
    nt_total += nt;

    const int wave_row0 = q0 + wrow * 32;
    const int my_row = wave_row0 + qi;
    const int pos_lo = rmod > 0 ? 0 : wave_row0, pos_hi = rmod > 0 ? rmod - 1 : wave_row0 + 31;   // positions of the wave's rows: [pos_lo, pos_hi]
    const int my_pos = rmod > 0 ? my_row % rmod : my_row;
    const int last_g = CAUSAL ? ((pos_hi + shift) >= 0 ? (pos_hi + shift) / BN : -1) : (ntg - 1);   // last tile of the head the wave's rows see
    const int wave_last_tile = KSPLIT ? (last_g >= grp ? (last_g - grp) >> 1 : -1) : last_g;

    float l4[4] = {0.f, 0.f, 0.f, 0.f};              // row sum of P, four interleaved partial sums carried across the tiles

    auto k_frag = [&](int kbuf, int i) -> X8 {   // fragment of QK^T MFMA i: key block i/DS, k-slot i%DS
      typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
      const unsigned a = (k_rd_addr ^ (KS(i) << 5)) + kbuf * TILE_BYTES + KT(i) * 32 * (D * 2);
      return __builtin_bit_cast(X8, *reinterpret_cast<lds_u32x4*>(a));
    };
    auto k_frag_rt = [&](unsigned kb_bytes, int i) -> X8 {   // same, K buffer chosen at run time
      typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
      const unsigned a = (k_rd_addr ^ (KS(i) << 5)) + kb_bytes + KT(i) * 32 * (D * 2);
      return __builtin_bit_cast(X8, *reinterpret_cast<lds_u32x4*>(a));
    };
    auto v_frag = [&](const char* vb, int i) -> X8 {   // fragment of PV MFMA i: key slot i/DT, d tile i%DT
      const char* a = vb + v_rd_base + ((i / DT) * 2 * DT << 9) + ((i % DT) << 9);
      s16x4 lo = lds_read_tr16_b64(a);
      s16x4 hh = lds_read_tr16_b64(a + 256);
      return __builtin_bit_cast(X8, __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto needs_mask = [&](int t) -> bool {
      const int key0 = key0_of(t);
      bool nm = (key0 + BN > p.Nk);
      if (CAUSAL) nm = nm || (key0 + BN - 1 > pos_lo + shift);
      return nm;
    };
    auto apply_mask = [&](int t, f32x16 (&s)[2]) {
      int lim = p.Nk - 1;
      if (CAUSAL) { const int c = my_pos + shift; lim = c < lim ? c : lim; }
      lim -= key0_of(t) + 4 * hi;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * tt + (r & 3) + 8 * (r >> 2);
          if (ko > lim) s[tt][r] = -INFINITY;
        }
    };
    // Reference exponent of the row ("mref", log2 domain) instead of the exact running max: P = exp2(s*sc - mref)
    // where mref is the row's scaled running max as of the last re-base.  A wave re-bases (every row takes its current
    // running max, O and l are multiplied by exp2(old - new)) only when some row max has outgrown its mref by more
    // than 2^8, so P <= 2^8: nothing can overflow in the fp32 sums or the 16-bit P, and O / l and the LSE are
    // unchanged mathematically.  That takes the O rescale (64 VALU per tile on ~60% of the tiles at N=4096 with the
    // exact-max rule, since SOME row of a wave nearly always sees a new max) out of the steady state; it lives in
    // the slow path below.  oracle/oracle.py:tiled_emulation_lazy restates this rule for the parity tests.
    float mref = -1e30f;
    auto trigger = [&](float mloc) -> bool { return __any(mloc * sc > mref + 8.f); };
    // (mloc may be the max over only this half-wave's 32 keys of the tile: the trigger is an OR over all lanes anyway;
    //  the re-base itself combines the two halves first so that both lanes of a row keep the same reference)
    auto rescale_if_needed = [&](float mloc) {
      if (__any(mloc * sc > mref + 8.f)) {
        const float x = pair_max(mloc) * sc;
        const float nref = fmaxf(mref, x);
        const float alpha = fast_exp2(mref - nref);
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i] *= alpha;
        o_scale<DT>(alpha);
        mref = nref;
      }
    };

    // one softmax element of the current tile: e in [0,32): P slot e>>3 (16 keys), position e&7.  The empty asm
    // statements pin each result inside the MFMA slot it was written in (IR passes otherwise re-associate the row
    // sum into packed adds and sink the row max to the end of the block, keeping all 32 exponentials live).
    typedef __attribute__((ext_vector_type(2))) T t2;
    auto soft_elem = [&](int e, const f32x16 (&s)[2], float msc, float (&lsum)[4], unsigned (&pw)[16], float& ev_hold) {
      const int slot = e >> 3, t = slot >> 1, r = (slot & 1) * 8 + (e & 7);
      const float ev = fast_exp2(fmaf(s[t][r], sc, -msc));
      lsum[e & 3] += ev;
      asm volatile("" : "+v"(lsum[e & 3]));
      if (e & 1) {
        const t2 w = {(T)ev_hold, (T)ev};
        pw[e >> 1] = __builtin_bit_cast(unsigned, w);
        asm volatile("" : "+v"(pw[e >> 1]));
      } else {
        ev_hold = ev;
      }
    };
    // The same element split into three stages that the fast path issues in three DIFFERENT MFMA slots (fma two slots
    // ahead, exp2 one slot ahead, sum/pack in the element's own slot): a wave issues in order, so a chain
    // fma -> exp -> add inside one slot would stall on every result; staged, all VALU work of a slot is independent.
    auto st_fma = [&](int e, const f32x16 (&s)[2], float msc, float (&xs)[32]) {
      const int slot = e >> 3, t = slot >> 1, r = (slot & 1) * 8 + (e & 7);
      xs[e] = fmaf(s[t][r], sc, -msc);
      asm volatile("" : "+v"(xs[e]));
    };
    auto st_exp = [&](int e, float (&xs)[32]) {
      xs[e] = fast_exp2(xs[e]);
      asm volatile("" : "+v"(xs[e]));
    };
    auto st_sum = [&](int e, float (&xs)[32], float (&lsum)[4], unsigned (&pw)[16]) {
      lsum[e & 3] += xs[e];
      asm volatile("" : "+v"(lsum[e & 3]));
      if (e & 1) {
        const t2 w = {(T)xs[e - 1], (T)xs[e]};
        pw[e >> 1] = __builtin_bit_cast(unsigned, w);
        asm volatile("" : "+v"(pw[e >> 1]));
      }
    };
    auto p_frag = [&](const unsigned (&pw)[16], int slot) -> X8 {
      const u32x4 w = {pw[4 * slot], pw[4 * slot + 1], pw[4 * slot + 2], pw[4 * slot + 3]};
      return __builtin_bit_cast(X8, w);
    };

    // ---- prologue: K(0), V(0), K(1) by DMA, Q fragments, S(0) and its row max -------------------------
    if (SEAM && seam_in) { /* the previous pass streamed K(0), K(1), V(0) and asked for Q */ }
    else if (!PREF || pass == 0) issue_prologue(mb, true);      // (with PREF the previous pass already asked for this block)
    // this pass continues its tile stream into the next one when there is one and the buffer parities line up (nt even)
    const bool seam = SEAM && (pass + 1 < npass) && ((nt & 1) == 0) && nt >= 2;
    o_zero<DT>();

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));
    asm volatile("s_barrier" ::: "memory");
    if (p.trace && pass == tr_pass) t_pro = __builtin_amdgcn_s_memtime();

    // tiles this wave computes: 0 .. nact-1 (causal: the waves of a block stop at different tiles)
    const int nact = (wave_last_tile + 1 < nt_own) ? (wave_last_tile + 1) : nt_own;
    // A wave whose 32 rows all lie behind the last query row computes nothing: decode-like problems have one valid wave per
    // block, and the tile rate — hence the K/V streaming rate — is then that one wave's.  (A separate flag, not a smaller
    // nact: non-causal kernels keep nact == nt as a compile-time fact and lose 2 % without it.)
    const bool idle_wave = (VF & VF_IL_IDLE) != 0 && wave_row0 >= p.Nq;
    // first tile of this wave that needs masking (causal diagonal or ragged tail); nact if none
    int fm = nact;
    {
      int first_g = 0x3fffffff;                            // first tile OF THE HEAD that needs a mask for this wave
      if (p.Nk % BN) first_g = p.Nk / BN;
      if (CAUSAL) {
        const int c = pos_lo + shift + 1;                  // keys 0..c-1 are visible to every row of the wave
        const int full = c > 0 ? c / BN : 0;               // tiles 0..full-1 need no mask
        first_g = full < first_g ? full : first_g;
      }
      const int first = KSPLIT ? (first_g <= grp ? 0 : (first_g - grp + 1) >> 1) : first_g;   // ... in the wave's own tile numbering
      fm = first < fm ? first : fm;
    }

    f32x16 sA[2], sB[2];
    float mA = -INFINITY, mB = -INFINITY;
    auto qk_burst = [&](int kbuf, int t, f32x16 (&s)[2], float& mout) {   // S(t) = K(t) Q^T from K buffer kbuf, masked, row max
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tt][r] = 0.f;
      const unsigned kb = kbuf * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < N1; ++i) s[KT(i)] = E::mfma(k_frag_rt(kb, i), qf[KS(i)], s[KT(i)]);
      if (needs_mask(t)) apply_mask(t, s);
      float mx = s[0][0];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tt][r]);
      mout = pair_max(mx);
    };
    if (nact > 0 && !idle_wave) {
      qk_burst(0, 0, sA, mA);
      mref = fmaxf(mref, mA * sc);                       // first re-base for free: O = 0 and l = 0 so far
    }
    // K buffer 0 is refilled with K(2) at the top of iteration 0: every wave must be done with K(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#if defined(TFA_IL_TRACEWAIT)
    // debug build: split the end of an iteration into (memory wait) and (barrier) and sum the cycles of each
    auto iter_end = [&]() {
      const unsigned long long a0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long a1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_barrier" ::: "memory");
      const unsigned long long a2 = __builtin_amdgcn_s_memtime();
      tw_wait += a1 - a0;
      tw_bar += a2 - a1;
    };
#else
    auto iter_end = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#endif

    // ---- fast path: tile j (S in scur) -> O; S(j+1) and its row max -> snext, mnext.  One basic block. ---------
    // PAR = j & 1: K(j+1) is in K buffer PAR^1, V(j) in V buffer PAR; K(j+2) -> K buffer PAR, V(j+1) -> V buffer PAR^1.
    auto fused = [&](auto par_c, auto mask_c, int j, f32x16 (&scur)[2], f32x16 (&snext)[2], float& mnext) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr bool MASK = decltype(mask_c)::value;   // tile j+1 is the wave's masked (diagonal / ragged) tile
      constexpr bool SPREAD = (VF & VF_IL_DMASPREAD) != 0;
      constexpr bool STAGGER = SPREAD && (VF & VF_IL_DMASTAGGER) != 0;
      const bool late = STAGGER && wave >= NW / 2;
      const bool issue_k = (j + 2 < nt) || seam;         // beyond the last tile: the next pass's K(0) / K(1)
      const int tk = (j + 2 < nt) ? j + 2 : j + 2 - nt;
      if (!SPREAD && !(AB & ILAB_NODMA)) {
        if (issue_k) dma_k(tk, PAR);
        dma_v(j + 1, PAR ^ 1);
      }
      const float msc = mref;
      constexpr int KB = PAR ^ 1;
      const char* vbp = vl + PAR * TILE_BYTES;
      unsigned pw[16];
      float xs[32];
      X8 kf[N1], vf[N2];
#pragma unroll
      for (int i = 0; i < PFK; ++i) kf[i] = k_frag(KB, i);
      // global MFMA slot g = 0..N1+N2-1; element e is summed/packed in slot SC(e), exponentiated in SC(e)-1, scaled in SC(e)-2
      auto soft_slot = [&](int g) {
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          const int sc_e = slot_of_elem(e);
          const int sb_e = sc_e - 1, sa_e = sc_e >= 2 ? sc_e - 2 : 0;
          if (AB & ILAB_NOEXP) {
            if (sc_e == g && (e & 1)) pw[e >> 1] = __builtin_bit_cast(unsigned, scur[e >> 4][e & 15]);
          } else {
            if (sa_e == g) st_fma(e, scur, msc, xs);
            if (sb_e == g) st_exp(e, xs);
            if (sc_e == g) st_sum(e, xs, l4, pw);
          }
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      // part 1
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        if (i + PFK < N1) kf[i + PFK] = ((AB & ILAB_NOKREAD) && i + PFK >= PFK) ? kf[(i + PFK) % PFK] : k_frag(KB, i + PFK);
        else vf[i + PFK - N1] = v_frag(vbp, i + PFK - N1);
        if (AB & ILAB_NOQK) {
          if (i < 2) asm volatile("" : "+v"(snext[i]));
        } else if (KS(i) == 0) {                           // first k-slot of a key block: C = 0 (inline constant)
#if TFA_IL_ASMQK
          s_mfma0<T>(snext[KT(i)], kf[i], qf[KS(i)]);
#else
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          snext[KT(i)] = E::mfma(kf[i], qf[KS(i)], z);
#endif
        } else {
#if TFA_IL_ASMQK
          s_mfma<T>(snext[KT(i)], kf[i], qf[KS(i)]);
#else
          snext[KT(i)] = E::mfma(kf[i], qf[KS(i)], snext[KT(i)]);
#endif
        }
        if (SPREAD && !(AB & ILAB_NODMA) && !(STAGGER && late)) {   // 2*PPW DMA pieces spread over the first MFMAs, one per MFMA
          if (i < PPW) dma_v1(j + 1, PAR ^ 1, i);
          else if (i < 2 * PPW) { if (issue_k) dma_k1(tk, PAR, i - PPW); }
        }
        soft_slot(i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MASK) {                                     // S(j+1) is complete (MFMA results: the s_nop covers the read distance)
#if TFA_IL_ASMQK
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
        apply_mask(j + 1, snext);
        __builtin_amdgcn_sched_barrier(0);
      }
      // part 2
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        if (i + PFV < N2) vf[i + PFV] = (AB & ILAB_NOVREAD) ? vf[(i + PFV) % PFV] : v_frag(vbp, i + PFV);
        if (!(AB & ILAB_NOPV)) o_mfma_d<T>(i % DT, vf[i], p_frag(pw, i / DT));
        else asm volatile("" ::"v"(vf[i]), "v"(pw[(i / DT) * 4]), "v"(pw[(i / DT) * 4 + 1]), "v"(pw[(i / DT) * 4 + 2]), "v"(pw[(i / DT) * 4 + 3]));
        if (STAGGER && late && !(AB & ILAB_NODMA)) {
          if (i < PPW) dma_v1(j + 1, PAR ^ 1, i);
          else if (i < 2 * PPW) { if (issue_k) dma_k1(tk, PAR, i - PPW); }
        }
        soft_slot(N1 + i);
#pragma unroll
        for (int q = 0; q < 16; ++q)               // 16 pairs of S(j+1) values -> one v_max3 each
          if (q * N2 / 16 == i && !(AB & ILAB_NOMAX)) {
            mx = fmaxf(fmaxf(mx, snext[q >> 3][2 * (q & 7)]), snext[q >> 3][2 * (q & 7) + 1]);
            asm volatile("" : "+v"(mx));
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      mnext = (AB & ILAB_NOMAX) ? snext[0][0] * 1e-30f : mx;   // this half-wave's 32 keys only (see rescale_if_needed)
      if (AB & ILAB_NOBARRIER) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else iter_end();
    };
    // ---- slow path: any tile (masked, last, re-base needed).  S(j) is in scur, S(j+1) goes to snext ------------------
    auto slow = [&](int j, f32x16 (&scur)[2], float mcur, f32x16 (&snext)[2], float& mnext) {
      const int par = j & 1;
      ++n_slow;
      if (j + 2 < nt) dma_k(j + 2, par);
      else if (seam) dma_k(j + 2 - nt, par);
      if (j + 1 < nt) dma_v(j + 1, par ^ 1);
      else if (seam) dma_v(0, par ^ 1);
      rescale_if_needed(mcur);
      const float msc = mref;
      const char* vbp = vl + par * TILE_BYTES;
      unsigned pw[16];
      float ev_hold = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) soft_elem(e, scur, msc, l4, pw, ev_hold);
#pragma unroll
      for (int i = 0; i < N2; ++i) o_mfma_d<T>(i % DT, v_frag(vbp, i), p_frag(pw, i / DT));
      if (j + 1 < nact) qk_burst(par ^ 1, j + 1, snext, mnext);
      iter_end();
    };

    // S(j) lives in sA for even j and in sB for odd j, on both paths, so the paths alternate freely without copies.
    // Tile j takes the fast path when tile j+1 exists and no row max of tile j has outgrown mref; if tile j+1 needs
    // masking (at most the wave's last one or two tiles) the body with the mask between its two parts runs.
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using MN = std::integral_constant<bool, false>;
    using MY = std::integral_constant<bool, true>;
    if (!idle_wave)
#pragma nounroll
    for (int j = 0; j < nact; j += 2) {
      if (j + 1 < nact && !trigger(mA)) {
        if (j + 1 < fm) fused(C0{}, MN{}, j, sA, sB, mB);
        else fused(C0{}, MY{}, j, sA, sB, mB);
      } else {
        slow(j, sA, mA, sB, mB);
      }
      if (j + 1 >= nact) break;
      if (j + 2 < nact && !trigger(mB)) {
        if (j + 2 < fm) fused(C1{}, MN{}, j + 1, sB, sA, mA);
        else fused(C1{}, MY{}, j + 1, sB, sA, mA);
      } else {
        slow(j + 1, sB, mB, sA, mA);
      }
    }
#pragma nounroll
    for (int j = idle_wave ? 0 : nact; j < nt; ++j) {    // tiles of the block this wave does not touch
      if (j + 2 < nt) dma_k(j + 2, j & 1);
      else if (seam) dma_k(j + 2 - nt, j & 1);
      if (j + 1 < nt) dma_v(j + 1, (j & 1) ^ 1);
      else if (seam) dma_v(0, (j & 1) ^ 1);
      iter_end();
    }
    if (p.trace && pass == tr_pass) t_loop = __builtin_amdgcn_s_memtime();

    // ---- epilogue ---------------------------------------------------------------------------
    // every wave is past the last tile's barrier: the K/V buffers and qf are free -> ask for the next pass's first tiles
    // and Q now, so that their latency hides behind the normalisation and the stores below
    if (PREF && pass + 1 < npass) issue_prologue(block_of(pass + 1), true);
    if (SEAM) { seam_in = seam; if (seam) issue_prologue(block_of(pass + 1), false); }   // Q only: K(0), K(1), V(0) are on chip
    float l_tot = pair_sum((l4[0] + l4[1]) + (l4[2] + l4[3]));
    float w0 = 1.f, w1 = 0.f;                          // KSPLIT: weights of this group's and the other group's partial result
    const char* const dump = smem + 4 * TILE_BYTES;    // KSPLIT: group 1's (now idle) tile buffers carry its O to group 0
    if constexpr (KSPLIT) {
      // the two groups hold partial results over disjoint key sets: group 1 writes (m, l, O) — lane-private data, no
      // transposition — and leaves; group 0 merges by the split-KV rule (tfa_merge.hip) while it reads its own O out
      float* const ml = reinterpret_cast<float*>(smem + wave * (32 * D * 2)) + lane * 2;   // (in group 0's idle buffers)
      if (grp == 1) {
        ml[0] = mref; ml[1] = l_tot;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          float o[16];
          if (d == 0) o_read<0>(o, 1.f); else if (d == 1) o_read<1>(o, 1.f); else if (d == 2) o_read<2>(o, 1.f); else o_read<3>(o, 1.f);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v4 = {o[4 * g + 0], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            *reinterpret_cast<f32x4*>(const_cast<char*>(dump) + ((((wave * DT + d) << 2) + g) << 10) + lane * 16) = v4;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (grp == 1) {
        if (pass + 1 >= npass) return;
        // paired causal blocks: group 0 still reads this group's buffers (the dump) and writes its own (epilogue slices);
        // the next pass's first DMA pieces wait behind the barrier group 0 ends its epilogue with
        asm volatile("s_barrier" ::: "memory");
        continue;
      }
      const float m1 = ml[0], l1 = ml[1];
      const float m = fmaxf(mref, m1);
      w0 = (mref == -INFINITY) ? 0.f : fast_exp2(mref - m);
      w1 = (m1 == -INFINITY) ? 0.f : fast_exp2(m1 - m);
      l_tot = w0 * l_tot + w1 * l1;
      mref = m;
    }
    const bool empty = !(l_tot > 0.f);
    const float inv = empty ? 1.f : 1.f / l_tot;
    // O[d tile d][0..15] of this lane, normalised (KSPLIT: merged with the other group's)
    auto o_get = [&](int d, float (&o)[16]) {
      const float f0 = inv * w0;
      if (d == 0) o_read<0>(o, f0); else if (d == 1) o_read<1>(o, f0); else if (d == 2) o_read<2>(o, f0); else o_read<3>(o, f0);
      if constexpr (KSPLIT) {
        const float f1 = inv * w1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(dump + ((((wave * DT + d) << 2) + g) << 10) + lane * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[4 * g + e] = fmaf(f1, x[e], o[4 * g + e]);
        }
      }
    };
    if (p.lse != nullptr && hi == 0 && my_row < p.Nq) {
      const float lse = empty ? INFINITY : (mref + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
      p.lse[(long long)bh * p.Nq + my_row] = lse;
    }
    if (F32OUT) {
      float* obase = reinterpret_cast<float*>(p.o) + b * p.os_b + h * p.os_h;
      auto o_rs = rsrc_at(obase, p.o_bytes, WIN ? (unsigned long long)q0 * (unsigned long long)p.os_n * 4ull : 0ull);
      const int ooff = (my_row - (WIN ? q0 : 0)) * (int)p.os_n * 4 + hi * 16;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        float o[16];
        o_get(d, o);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v4 = {o[4 * g + 0], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), o_rs, d * 32 + g * 8 + hi * 4 < p.dv ? ooff + (d * 32 + g * 8) * 4 : (int)TFA_OOB, 0, 0);
        }
      }
    } else if (EPI) {
      // A lane holds 4-element pieces of ONE row spread over 16 register groups: stored directly that is 16 eight-byte
      // stores per lane, 32 different rows per instruction.  Instead the wave transposes its 32 x D tile through its own
      // slice of the epilogue region (16-byte chunk index XOR row, as for K) and writes whole rows: 1 KiB contiguous per
      // store instruction.  The region is separate from the tile buffers (which the next pass is already filling).
      T* obase = reinterpret_cast<T*>(p.o) + b * p.os_b + h * p.os_h;
      auto o_rs = rsrc_at(obase, p.o_bytes, WIN ? (unsigned long long)q0 * (unsigned long long)p.os_n * 2ull : 0ull);
      typedef __attribute__((ext_vector_type(4))) T t4;
      // (the lane ids go through an empty asm so that none of the 24 addresses below is loop-invariant: hoisted out of
      // the pass loop they would stay live across the main loop and spill)
      int qix = qi, lanex = lane;
      asm volatile("" : "+v"(qix), "+v"(lanex));
      constexpr bool INPLACE = (VF & VF_IL_EPI_INPLACE) != 0;
      static_assert(!INPLACE || !PREF, "in-place epilogue slices would be overwritten by the next pass's prefetch");
      char* const ow = smem + (INPLACE ? 0 : 4 * TILE_BYTES) + wave * (32 * D * 2);
      constexpr int CH = D / 8;                      // 16-byte chunks per row
      const int osw = (CH == 16) ? (qix & 15) : (qix & 7);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        float o[16];
        o_get(d, o);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          t4 v4 = {(T)o[4 * g + 0], (T)o[4 * g + 1], (T)o[4 * g + 2], (T)o[4 * g + 3]};
          const int c = d * 4 + g;
          *reinterpret_cast<u32x2*>(ow + qix * (D * 2) + ((c ^ osw) << 4) + (lanex >> 5) * 8) = __builtin_bit_cast(u32x2, v4);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no barrier needed
      constexpr int RPI = 64 / CH;                   // rows per store instruction (4 at D=128, 8 at D=64)
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + lanex / CH, cpos = lanex % CH;
        const int c = cpos ^ ((CH == 16) ? (r & 15) : (r & 7));
        u32x4 v = *reinterpret_cast<const u32x4*>(ow + r * (D * 2) + (cpos << 4));
#ifndef TFA_IL_OSTORE_AUX
#define TFA_IL_OSTORE_AUX 2     // cache policy of the O row stores: nt — O is written once and never re-read, the XCD's L2 is better spent on
                                // the K/V tiles every query block of the head re-reads (cfg3: +1.8 %, others +-0; profiles/r02_ostore_ab.txt)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, c * 8 < p.dv ? (wave_row0 - (WIN ? q0 : 0) + r) * (int)p.os_n * 2 + (c << 4) : (int)TFA_OOB, 0, TFA_IL_OSTORE_AUX);
      }
      if (INPLACE) {
        // the next pass's first DMA pieces land in these buffers: every wave must have read its rows back
        if (!KSPLIT && pass + 1 < npass) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (KSPLIT: below, every output type)
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is rewritten by this wave's next epilogue only
      }
    } else {
      T* obase = reinterpret_cast<T*>(p.o) + b * p.os_b + h * p.os_h;
      auto o_rs = rsrc_at(obase, p.o_bytes, WIN ? (unsigned long long)q0 * (unsigned long long)p.os_n * 2ull : 0ull);
      const int ooff = (my_row - (WIN ? q0 : 0)) * (int)p.os_n * 2 + hi * 8;
      typedef __attribute__((ext_vector_type(4))) T t4;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        float o[16];
        o_get(d, o);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          t4 v4 = {(T)o[4 * g + 0], (T)o[4 * g + 1], (T)o[4 * g + 2], (T)o[4 * g + 3]};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v4), o_rs, d * 32 + g * 8 + hi * 4 < p.dv ? ooff + (d * 32 + g * 8) * 2 : (int)TFA_OOB, 0, 0);
        }
      }
    }
    // KSPLIT, another pass to come: group 0 has read group 1's dump and its own epilogue slices — both live in tile buffers the
    // next pass's first DMA pieces overwrite (group 1 waits at the matching barrier right behind the merge)
    if (KSPLIT && pass + 1 < npass) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  if (p.trace) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
      t[0] = t_start; t[1] = t_pro; t[2] = t_loop; t[3] = t_end;
#if defined(TFA_IL_TRACEWAIT)
      t[1] = t_start + tw_wait; t[2] = t_start + tw_wait + tw_bar;   // debug: "prologue" = memory waits, "loop" = barrier waits of wave 0
#endif
      t[4] = (unsigned long long)nt_total | ((unsigned long long)n_slow << 32);   // wave 0's slow-path tiles in the high half
      t[5] = (unsigned long long)__builtin_amdgcn_s_getreg(63508) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32);   // XCC_ID | HW_ID << 32
      t[6] = __builtin_amdgcn_s_memrealtime() - rt_start;   // 100 MHz ticks over the same span as t[3] - t[0] shader cycles
      t[7] = ((unsigned long long)bh << 32) | (unsigned)wi;
    }
  }
}

#undef KT
#undef KS

}  // namespace tfa
