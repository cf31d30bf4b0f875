// tfa_fwd_kernel_x4.h — issue-interleaved forward kernel, 4 waves x 64 query rows: ONE wave per SIMD, the whole
// 512-entry register file per wave (gfx950).
//
// Why (profiles/r01_pmc_default_cfg3.txt, DESIGN.md "Where the time goes"): in the 8-wave il kernel the two waves of a
// SIMD time-slice (the issue arbiter serves the older wave first), every wave re-reads the whole K and V tile from LDS
// (one fragment read per MFMA), and 770 of 2920 cycles per tile sit at the barrier.  Here a wave owns TWO 32-row query
// blocks, so
//   * every K / V fragment read from LDS feeds two MFMAs (LDS fragment traffic per flop halves, and so does the
//     instruction count of the reads);
//   * one instruction stream per SIMD: nothing to arbitrate, the barrier joins 4 waves that run in lockstep on 4 SIMDs;
//   * the 64 MFMAs of a tile (32 S = K Q^T, 32 O += P V) form one software-pipelined stream in which each MFMA is
//     followed by ~1 softmax element of the PREVIOUS tile (scale/subtract two slots ahead, exp2 one slot ahead,
//     sum + 16-bit pack in its own slot), one fragment read, and in the PV half one v_max3 of the next tile's row max —
//     the "<= 5 single-issue fillers per MFMA gap" budget of /opt/skills/guides/MI355X_MICROARCH.md.
// Registers per lane: O 2x4x16 = 128 and the Q fragments 2x8x4 = 64 live in hand-owned AGPRs a[0:191] (named by every
// asm that touches them, never by the compiler); hipcc's allocator owns only the 256 architectural VGPRs:
// S(j) 64, S(j+1) 64, P 32, fragments in flight ~24, row state ~20.
// K tiles sit in a ring of THREE LDS buffers (V: two) so that the first fragments of K(j+2) are read in the tail of
// iteration j, before the barrier: the first MFMA after the barrier has its operands in registers.
// Numerics: the lazily re-based row reference of tfa_fwd_kernel_il.h, per 32-row block (same rule, same emulation:
// oracle.tiled_emulation_lazy(group=32, thresh=8)).
#pragma once
#include <type_traits>
#include "tfa_fwd_kernel_dma.h"
#include "tfa_fwd_x4_asm_loop.inc"
#if !defined(TFA_X4_USE_ASMLOOP)
#define TFA_X4_USE_ASMLOOP 1     // 0: the compiler-scheduled tile body everywhere (the A/B arm of the hand-scheduled steady state)
#endif

namespace tfa {

constexpr int VF_X4 = 1 << 22;            // this kernel
constexpr int VF_X4_EPI = 1 << 23;        // 16-bit O leaves through a separate LDS region as whole rows (16-byte stores)
constexpr int VF_X4_WINDOWED = 1 << 24;    // (b,h) slices of 2 GiB and more: K/V through one descriptor per tile, Q / O through one per query block (rsrc_at), as the il
                                           // kernels' VF_IL_WINDOWED; the compiler-scheduled tile bodies only (the hand-scheduled loop adds the tile offset as the DMA's scalar offset)
constexpr int VF_X4_EPI_INPLACE = 1 << 30; // ... through the (idle) tile buffers instead: D = 256, whose five 32 KiB buffers fill the LDS.  A barrier
                                          // behind the epilogue when another pass follows (its first DMA pieces land in other waves' slices)

// ---- hand-owned accumulator registers -------------------------------------------------------------------------
// O (row block rb, d tile d) lives in a[(rb*DT + d)*16 .. +15], the Q fragment (rb, k-slot ks) in a[128 + (rb*DS + ks)*4 .. +3].
// They are NOT compiler values: with "+a" operands hipcc's allocator kept part of Q in VGPRs and scratch and re-loaded it
// in front of every S MFMA (596 spilled registers).  Every access is inline asm naming the physical registers, and every
// such asm — the S MFMAs too, which only READ Q — lists ALL of a0..a191 as clobbered: that makes the kernel descriptor
// allocate them and tells the allocator that nothing of its own survives there across any of these statements (it parks
// values in AGPRs around high-pressure regions: with Q unclobbered by the S MFMAs it chose a128.. for loop invariants, with
// Q unclobbered by the PV MFMAs it parked the prefetched K fragments in a128..a131 across the slow path's PV burst).  The compiler's own
// AGPR use must stay in a192..a255: tests/test_abi.py disassembles the library and fails on any v_accvgpr_* / AGPR operand
// below a192 outside these asm statements.
#define TFA_X4_OCLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define TFA_X4_QCLOB "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"
#define TFA_X4_OLIST "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127"
#define TFA_X4_ALLCLOB TFA_X4_OCLOB, TFA_X4_QCLOB
constexpr int X4_QBASE = 128;

// MFMA forms.  S^T = K Q^T: A = K fragment (VGPR, fresh from LDS), B = Q fragment in its AGPRs, C/D = S in VGPRs.
// O^T += V^T P^T: A = V fragment, B = P fragment (VGPRs), C/D = O in its AGPRs.
// Hazards the compiler cannot see inside an asm: (1) a VALU write of an A/B operand needs 2 wait states before the MFMA
// reads it — every P fragment is packed at least one whole MFMA slot before its first use; (2) an MFMA result needs
// the MFMA's passes before a non-MFMA instruction may read or write it — S is first touched by VALU code >= 2 MFMA
// issues after the last MFMA of its chain, anything sooner goes through x4_fence_v() / the s_nops of the O accessors.
template <typename T> struct X4 {
  using X8 = typename Elem<T>::x8;
  static constexpr bool bf = std::is_same<T, __bf16>::value;
  template <int QLO> static __device__ __forceinline__ void qk0(f32x16& c, X8 k) {
    if constexpr (bf) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(c) : "v"(k), "i"(QLO), "i"(QLO + 3) : TFA_X4_ALLCLOB);
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(c) : "v"(k), "i"(QLO), "i"(QLO + 3) : TFA_X4_ALLCLOB);
  }
  template <int QLO> static __device__ __forceinline__ void qk(f32x16& c, X8 k) {
    if constexpr (bf) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(c) : "v"(k), "i"(QLO), "i"(QLO + 3) : TFA_X4_ALLCLOB);
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(c) : "v"(k), "i"(QLO), "i"(QLO + 3) : TFA_X4_ALLCLOB);
  }
  template <int OLO> static __device__ __forceinline__ void pv(X8 v, X8 pfrag) {
    if constexpr (bf) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OLO), "i"(OLO + 15) : TFA_X4_ALLCLOB);
    else asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OLO), "i"(OLO + 15) : TFA_X4_ALLCLOB);
  }
};
static __device__ __forceinline__ void x4_fence_v(f32x16& c) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c)); }
// one 32-bit word of a Q fragment into its AGPR
template <int R> static __device__ __forceinline__ void x4_q_write(unsigned w) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(w), "i"(R) : TFA_X4_ALLCLOB);
}
static __device__ __forceinline__ void x4_o_zero() {
  asm volatile(".irp r," TFA_X4_OLIST "\n\tv_accvgpr_write_b32 a[\\r], 0\n\t.endr" ::: TFA_X4_ALLCLOB);
}
// every pending MFMA result has left the pipe (before a non-MFMA instruction touches O)
static __device__ __forceinline__ void x4_o_fence() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: TFA_X4_ALLCLOB); }
template <int R> static __device__ __forceinline__ float x4_o_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
  return x;
}
template <int R> static __device__ __forceinline__ void x4_o_write(float x) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(x), "i"(R) : TFA_X4_ALLCLOB);
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N-1.  The fast path below is written with these (and
// `if constexpr`) instead of `#pragma unroll` + runtime ifs: 64 MFMA slots x 64 softmax elements of compare-and-select is
// more than LLVM's full-unroll budget, and a loop left rolled turns every register array into scratch memory.
template <int I, int N, typename F> static __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// MFMA slot (0 .. N1+N2-1) in which softmax element n (0 .. NE-1, in order of need) is summed and packed
template <int N1, int NE1, int NE, int TAIL> static constexpr int x4_slot_of(int n) {
  return 1 + (n < NE1 ? n * N1 / NE1 : N1 - 1 + (n - NE1) * TAIL / (NE - NE1));
}

// AB: timing-only ablation bits of the fast path (results are wrong when set; tools/ablate_x4.py)
constexpr int X4AB_NOEXP = 1, X4AB_NODMA = 2, X4AB_NOBARRIER = 4, X4AB_NOMAX = 8, X4AB_NOQK = 16, X4AB_NOPV = 32, X4AB_NOKREAD = 64,
              X4AB_NOVREAD = 128;

#ifndef TFA_X4_NE1
#define TFA_X4_NE1 40
#endif
#ifndef TFA_X4_PF
#define TFA_X4_PF 2
#endif
// MFMA slot of DMA piece n (0 .. 2*PPW-1; V pieces first): first slot and stride.  Odd slots carry no LDS fragment read
// (those sit in front of the even MFMAs), and an LDS-DMA issued next to ds_reads costs 100-185 cycles of issue with one
// wave per SIMD against 25-60 in a VALU-only gap (MI355X_MICROARCH.md, per-instruction constants).
#ifndef TFA_X4_DMA0
#define TFA_X4_DMA0 1
#endif
#ifndef TFA_X4_DMASTEP
#define TFA_X4_DMASTEP 1
#endif

// RB = 32-row blocks per wave: 2 at D <= 128 (256-row workgroups), 1 at D = 256 (128-row workgroups) — RB * D = 256 keeps O
// (RB * D/32 * 16 = 128 registers) and Q (RB * D/16 * 4 = 64) in the same hand-owned AGPRs and a tile at 64 MFMAs per wave.
// DVB (256-wide form only): 32-wide column blocks that can hold valid head-dim columns, ceil(head dim / 32) = 5..8.  The LDS tiles and
// every address stay 256 wide (the columns beyond the head dim are the descriptors' zeros); K / Q fragments, V fragments, the MFMAs
// and the O registers cover the valid blocks only (head dim 192: 48 of 64 MFMAs per tile).
template <typename T, int D, bool CAUSAL, bool F32OUT, int VF, int AB = 0, int RB = (D <= 128 ? 2 : 1), int DVB = D / 32>
__global__ __launch_bounds__(256, 1) void fwd_kernel_x4(const KArgs p) {
  using M = X4<T>;
  using X8 = typename M::X8;
  constexpr int NW = 4;
  static_assert(RB * D <= 256 && (RB == 1 || RB == 2), "O and Q must fit a0..a191");
  constexpr int BM = NW * RB * 32;                 // 256 query rows per workgroup
  constexpr int BN = 64;
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NW;                 // DMA pieces per wave per tensor per tile
  static_assert(DVB == D / 32 || (D == 256 && RB == 1 && DVB >= 5 && DVB < 8), "DVB < D/32 exists for the 256-wide form only");
  constexpr int DS = 2 * DVB;                      // k-slots of 16 columns that are multiplied
  constexpr int DT = DVB;                          // 32-column tiles of O that exist
  constexpr int DT_L = D / 32;                     // ... as the V tile's LDS layout counts them
  constexpr int NKF = 2 * DS;                      // K fragments per tile (key block kt = i & 1, k-slot ks = i >> 1)
  constexpr int NVF = 4 * DT;                      // V fragments per tile (key slot s = i / DT, d tile d = i % DT)
  constexpr int N1 = RB * NKF;                     // S MFMAs per tile
  constexpr int N2 = RB * NVF;                     // PV MFMAs per tile
  constexpr int NE = RB * 32;                      // softmax elements per lane per tile
  constexpr int NE1 = TFA_X4_NE1 * RB / 2;         // of those, summed/packed during part 1
  constexpr int PF = TFA_X4_PF;                    // fragment read-ahead, in fragments (= 2 MFMAs each)
  constexpr int NKB = 3;                           // K ring
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  constexpr bool EPI_IN = (VF & VF_X4_EPI_INPLACE) != 0;
  constexpr bool WIN = (VF & VF_X4_WINDOWED) != 0;
  constexpr bool EPI = (VF & VF_X4_EPI) != 0 || EPI_IN;
  static_assert(!EPI_IN || 4 * RB * 32 * D * 2 <= (NKB + 2) * 64 * D * 2, "in-place epilogue slices inside the tile buffers");
  static_assert(PPW >= 1 && PPW * NW == PIECES, "tile does not split into whole DMA pieces per wave");
  static_assert(PF >= 1 && PF <= NKF && PF <= NVF, "");

  // Softmax element n (0..63) of a tile: P slot pair pp = n / 16 (16 keys), row block rb = (n % 16) / 8, position n % 8;
  // e = 8 * pp + n % 8 is its index among the row block's 32 elements.  The order is the order of need: PV MFMA
  // N1 + RB*DT*pp is the first to read P slot pp.  MFMA slot (0..N1+N2-1) in which element n is summed and packed:
#define X4_SLOT_OF(n) x4_slot_of<N1, NE1, NE, RB * DT * 3 - 1>(n)
  constexpr int EG = 8 * RB;                       // elements per P slot (all row blocks)
  static_assert(X4_SLOT_OF(EG - 1) < N1 && X4_SLOT_OF(2 * EG - 1) < N1 + RB * DT && X4_SLOT_OF(3 * EG - 1) < N1 + 2 * RB * DT &&
                    X4_SLOT_OF(4 * EG - 1) < N1 + 3 * RB * DT,
                "a P slot is packed too late for the PV MFMA that reads it");
  // element n -> row block and index among the row block's 32 elements
#define X4_EL_RB(n) (((n) % EG) >> 3)
#define X4_EL_E(n) (((n) / EG) * 8 + ((n) & 7))

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const vl = smem + NKB * TILE_BYTES;        // V buffers 0,1 behind the three K buffers
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, rt_start = 0;
  if (p.trace) { rt_start = __builtin_amdgcn_s_memrealtime(); t_start = __builtin_amdgcn_s_memtime(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31;
  const int hi = lane >> 5;

  int bh, wi;
  {
    const int id = blockIdx.x;
    if ((p.nbh & 7) == 0) {                        // a head stays on one XCD (block id & 7): its K/V is re-read out of that L2
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
  // one descriptor per slice (the host guarantees < 2 GiB) — or, WINDOWED, K/V: one per tile, Q / O: one per query block, offsets relative to it
  auto k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, WIN ? 0u : (unsigned)p.k_bytes, 0x00020000);
  auto v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, WIN ? 0u : (unsigned)p.v_bytes, 0x00020000);
  auto slice_rsrc = [&](const void* base, unsigned long long bytes, unsigned long long off) {
    if constexpr (WIN) return rsrc_at(base, bytes, off);
    else return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)bytes, 0x00020000);
  };

  // per-lane DMA source offsets of this wave's pieces (tile 0); the K swizzle and the V sub-tile order are applied to the
  // SOURCE address, the LDS destination of piece pc is pc * 1024 + lane * 16 (tfa_fwd_kernel_dma.h)
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
      const int dt = sub % DT_L, sh = sub / DT_L;
      const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
      v_src[i] = (dt * 4 + pcs) * 8 < p.dv ? key * (int)p.vs_n * 2 + ((dt * 4 + pcs) << 4) : (int)TFA_OOB;
    }
  }
  const int k_tile_stride = BN * (int)p.ks_n * 2;
  const int v_tile_stride = BN * (int)p.vs_n * 2;
  const unsigned my_piece0 = lds_base + wave * PPW * 1024;
  // piece i of K tile t into the ring buffer at kboff.  WINDOWED: the tile's own descriptor; `off` = the lane's source offset inside the tile, or TFA_OOB
  auto dma_k_win = [&](int t, unsigned kboff, int i, int off) {
    lds_dma16_m0_fresh(rsrc_at(kbase, p.k_bytes, (unsigned long long)(unsigned)t * (unsigned)k_tile_stride), my_piece0 + kboff + i * 1024, off);
  };
  auto dma_k1 = [&](int t, unsigned kboff, int i) {
    if constexpr (WIN) dma_k_win(t, kboff, i, k_src[i]);
    else lds_dma16_m0(k_rs, my_piece0 + kboff + i * 1024, k_src[i] + t * k_tile_stride);
  };
  auto dma_v1 = [&](int t, int vbuf, int i) {
    if constexpr (WIN) lds_dma16_m0_fresh(rsrc_at(vbase, p.v_bytes, (unsigned long long)(unsigned)t * (unsigned)v_tile_stride), my_piece0 + (NKB + vbuf) * TILE_BYTES + i * 1024, v_src[i]);
    else lds_dma16_m0(v_rs, my_piece0 + (NKB + vbuf) * TILE_BYTES + i * 1024, v_src[i] + t * v_tile_stride);
  };
  auto dma_k = [&](int t, unsigned kboff) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_k1(t, kboff, i);
  };
  auto dma_v = [&](int t, int vbuf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_v1(t, vbuf, i);
  };

  // K fragment (kt, ks): 16-byte chunk (2*ks + hi) ^ swz of row qi of key block kt.  2*ks and hi share no bits, so the
  // address is (row base + ((hi ^ swz) << 4)) ^ (ks << 5) + buffer offset + kt * 32 rows (smem is 1 KiB aligned).
  unsigned k_rd_addr = lds_base + qi * (D * 2) + ((hi ^ k_swz<D>(qi)) << 4);
  asm volatile("" : "+v"(k_rd_addr));
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT_L << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const float sc = p.scale_log2;
  int nt_total = 0, n_slow = 0, n_trig = 0;
  const int dbg = p.dbg;                          // debug flags (tfa_debug_set_flags; 0 in normal use): see the uses below
  auto big_fence = [&]() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };

  typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
  auto k_frag = [&](unsigned kboff, int i) -> X8 {
    const unsigned a = ((k_rd_addr + kboff) ^ ((i >> 1) << 5)) + (i & 1) * 32 * (D * 2);
    return __builtin_bit_cast(X8, *reinterpret_cast<lds_u32x4*>(a));
  };
  auto v_frag = [&](const char* vb, int i) -> X8 {
    const char* a = vb + v_rd_base + ((i / DT) * 2 * DT_L << 9) + ((i % DT) << 9);
    s16x4 lo = lds_read_tr16_b64(a);
    s16x4 hh = lds_read_tr16_b64(a + 256);
    return __builtin_bit_cast(X8, __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  const int npass = PAIR ? ((p.nmb - 1 - wi) != wi ? 2 : 1) : 1;
  auto block_of = [&](int pass) -> int {
    if (PAIR) return pass == 0 ? (p.nmb - 1 - wi) : wi;
    return CAUSAL ? (p.nmb - 1 - wi) : wi;
  };

  const int tr_pass = (p.dbg & 128) ? 1 : 0;
#pragma nounroll
  for (int pass = 0; pass < npass; ++pass) {
    const int mb = block_of(pass);
    if (p.trace && pass == 1 && tr_pass == 1) t_start = __builtin_amdgcn_s_memtime();
    const int q0 = mb * BM;
    int kv_end = p.Nk;
    if (CAUSAL) {
      const int lim = q0 + BM + shift;
      kv_end = lim < kv_end ? lim : kv_end;
    }
    const int nt = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
    nt_total += nt;
    const int wave_row0 = q0 + wave * (32 * RB);

    // ---- requests: K(0), V(0), K(1), K(2) by LDS-DMA, this lane's Q fragments ------------------------------------
    if (nt > 0) dma_k(0, 0);
    if (nt > 0) dma_v(0, 0);
    if (nt > 1) dma_k(1, TILE_BYTES);
    if (nt > 2) dma_k(2, 2 * TILE_BYTES);
    {
      u32x4 qv[RB][DS];
      auto q_rs = slice_rsrc(qbase, p.q_bytes, (unsigned long long)(unsigned)q0 * (unsigned long long)p.qs_n * 2);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int qoff = (wave_row0 - (WIN ? q0 : 0) + rb * 32 + qi) * (int)p.qs_n * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < DS; ++s) qv[rb][s] = __builtin_amdgcn_raw_buffer_load_b128(q_rs, (2 * s + hi) * 8 < p.dv ? qoff + s * 32 : (int)TFA_OOB, 0, 0);
      }
      x4_o_fence();                                    // (second pass: the previous epilogue's reads of O are long done; cheap)
      x4_o_zero();
      static_for<0, RB * DS>([&](auto f_c) {            // Q fragment (rb, ks) -> a[128 + 4*(rb*DS + ks) ..]
        constexpr int f = decltype(f_c)::value;
        x4_q_write<X4_QBASE + 4 * f + 0>(qv[f / DS][f % DS][0]);
        x4_q_write<X4_QBASE + 4 * f + 1>(qv[f / DS][f % DS][1]);
        x4_q_write<X4_QBASE + 4 * f + 2>(qv[f / DS][f % DS][2]);
        x4_q_write<X4_QBASE + 4 * f + 3>(qv[f / DS][f % DS][3]);
      });
    }
    float l4[RB][4];                                 // row sum of P: four interleaved partial sums per row block, carried across tiles
    float mref[RB];                                  // reference exponent of the row (log2 domain), see tfa_fwd_kernel_il.h
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      mref[rb] = -1e30f;
#pragma unroll
      for (int i = 0; i < 4; ++i) l4[rb][i] = 0.f;
    }

    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (p.trace && pass == tr_pass) t_pro = __builtin_amdgcn_s_memtime();

    // tiles this wave computes: 0 .. nact-1 (causal: the waves of a block stop at different tiles)
    const int wave_last_tile = CAUSAL ? ((wave_row0 + 32 * RB - 1 + shift) >= 0 ? (wave_row0 + 32 * RB - 1 + shift) / BN : -1) : (nt - 1);
    const int nact = (wave_last_tile + 1 < nt) ? (wave_last_tile + 1) : nt;
    int fm = nact;                                   // first tile that needs masking for some row of the wave; nact if none
    {
      const int ragged = (p.Nk % BN) ? (p.Nk / BN) : nact;
      fm = ragged < fm ? ragged : fm;
      if (CAUSAL) {
        const int c = wave_row0 + shift + 1;         // keys 0..c-1 are visible to every row of the wave
        const int full = c > 0 ? c / BN : 0;
        fm = full < fm ? full : fm;
      }
    }

    auto apply_mask = [&](int t, int rb, f32x16 (&s)[2]) {
      int lim = p.Nk - 1;
      if (CAUSAL) { const int c = wave_row0 + rb * 32 + qi + shift; lim = c < lim ? c : lim; }
      lim -= t * BN + 4 * hi;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * tt + (r & 3) + 8 * (r >> 2);
          if (ko > lim) s[tt][r] = -INFINITY;
        }
    };
    auto needs_mask = [&](int t) -> bool {
      const int key0 = t * BN;
      bool nm = (key0 + BN > p.Nk);
      if (CAUSAL) nm = nm || (key0 + BN - 1 > wave_row0 + shift);
      return nm;
    };
    auto trigger = [&](const float (&m)[RB]) -> bool {
      bool t = m[0] * sc > mref[0] + 8.f;
      if constexpr (RB == 2) t = t || (m[1] * sc > mref[1] + 8.f);
      return __any(t);
    };
    // cold: row block rb takes its current running max as the new reference; O and l are multiplied by exp2(old - new)
    auto rescale_if_needed = [&](auto rb_c, float mloc) {
      constexpr int rb = decltype(rb_c)::value;
      if (__any(mloc * sc > mref[rb] + 8.f)) {
        const float x = pair_max(mloc) * sc;
        const float nref = fmaxf(mref[rb], x);
        const float alpha = fast_exp2(mref[rb] - nref);
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[rb][i] *= alpha;
        x4_o_fence();
        static_for<0, DT * 16>([&](auto r_c) {
          constexpr int R = rb * DT * 16 + decltype(r_c)::value;
          x4_o_write<R>(x4_o_read<R>() * alpha);
        });
        asm volatile("s_nop 3" ::: "memory");          // v_accvgpr_write -> MFMA reads it as C
        mref[rb] = nref;
      }
    };

    typedef __attribute__((ext_vector_type(2))) T t2;
    // Softmax element n of the current tile in three stages issued in three DIFFERENT MFMA slots (a wave issues in order:
    // fma -> exp -> add inside one slot would stall on every result).  The empty asm statements pin each result inside the
    // slot it was written in (IR passes otherwise re-associate the row sum into packed adds, tfa_fwd_kernel_il.h).
    auto st_fma = [&](auto n_c, const f32x16 (&s)[RB][2], float (&xs)[NE]) {
      constexpr int n = decltype(n_c)::value, rb = X4_EL_RB(n), e = X4_EL_E(n), slot = e >> 3, t = slot >> 1, r = (slot & 1) * 8 + (e & 7);
      xs[n] = fmaf(s[rb][t][r], sc, -mref[rb]);
      asm volatile("" : "+v"(xs[n]));
    };
    auto st_exp = [&](auto n_c, float (&xs)[NE]) {
      constexpr int n = decltype(n_c)::value;
      xs[n] = fast_exp2(xs[n]);
      asm volatile("" : "+v"(xs[n]));
    };
    auto st_sum = [&](auto n_c, float (&xs)[NE], unsigned (&pw)[RB][16]) {
      constexpr int n = decltype(n_c)::value, rb = X4_EL_RB(n), e = X4_EL_E(n);
      l4[rb][e & 3] += xs[n];
      asm volatile("" : "+v"(l4[rb][e & 3]));
      if constexpr (e & 1) {
        const t2 w = {(T)xs[n - 1], (T)xs[n]};
        pw[rb][e >> 1] = __builtin_bit_cast(unsigned, w);
        asm volatile("" : "+v"(pw[rb][e >> 1]));
      }
    };
    auto p_frag = [&](const unsigned (&pw)[RB][16], int rb, int slot) -> X8 {
      const u32x4 w = {pw[rb][4 * slot], pw[rb][4 * slot + 1], pw[rb][4 * slot + 2], pw[rb][4 * slot + 3]};
      return __builtin_bit_cast(X8, w);
    };

    f32x16 sA[RB][2], sB[RB][2];
    float mA[RB], mB[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) mA[rb] = mB[rb] = -INFINITY;
    X8 kpre[PF];                                     // first fragments of the next iteration's K tile, read before the barrier
    // S(t) = K(t) Q^T from the K buffer at kboff, masked, row max per row block (this half-wave's 32 keys only)
    auto qk_burst = [&](unsigned kboff, int t, f32x16 (&s)[RB][2], float (&mout)[RB]) {
      static_for<0, NKF>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        const X8 kf = k_frag(kboff, i);
        static_for<0, RB>([&](auto rb_c) {
          constexpr int rb = decltype(rb_c)::value;
          if constexpr ((i >> 1) == 0) M::template qk0<X4_QBASE + 4 * (rb * DS + (i >> 1))>(s[rb][i & 1], kf);
          else M::template qk<X4_QBASE + 4 * (rb * DS + (i >> 1))>(s[rb][i & 1], kf);
        });
      });
      if (dbg & 8) big_fence();
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        x4_fence_v(s[rb][0]);
        x4_fence_v(s[rb][1]);
        if (needs_mask(t)) apply_mask(t, rb, s[rb]);
        float mx = s[rb][0][0];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[rb][tt][r]);
        mout[rb] = mx;
      }
    };
    auto load_kpre = [&](unsigned kboff) {
#pragma unroll
      for (int i = 0; i < PF; ++i) kpre[i] = k_frag(kboff, i);
    };

    // K ring: byte offsets of the buffers of K(j), K(j+1), K(j+2) for the current iteration j
    unsigned kb0 = 0, kb1 = TILE_BYTES, kb2 = 2 * TILE_BYTES;
    auto rotate = [&]() { const unsigned t = kb0; kb0 = kb1; kb1 = kb2; kb2 = t; };

    if (nact > 0) {
      qk_burst(0, 0, sA, mA);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) mref[rb] = fmaxf(mref[rb], pair_max(mA[rb]) * sc);   // first re-base for free: O = 0, l = 0
      if (nact > 1) load_kpre(kb1);
    }
    // K buffer 0 is refilled with K(3) in iteration 0: every wave must be done with K(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto iter_end = [&]() {
      if (AB & X4AB_NOBARRIER) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // ---- fast path: tile j (S in scur) -> O; S(j+1) and its row max -> snext, mnext. ------------------------------
    // PAR = j & 1: V(j) is in V buffer PAR, V(j+1) -> V buffer PAR^1; K(j+1) at kb1 (first PF fragments already in
    // kpre), K(j+2) at kb2 (its first fragments are read in the tail), K(j+3) -> kb0.
    auto fused = [&](auto par_c, int j, f32x16 (&scur)[RB][2], f32x16 (&snext)[RB][2], float (&mnext)[RB]) {
      constexpr int PAR = decltype(par_c)::value;
      const bool issue_k = (j + 3 < nt);
      const char* vbp = vl + PAR * TILE_BYTES;
      unsigned pw[RB][16];
      float xs[NE];
      X8 kf[NKF], vf[NVF];
#pragma unroll
      for (int i = 0; i < PF; ++i) kf[i] = kpre[i];
      // The prefetched fragments are loop-carried compiler values: under register pressure hipcc parks them in AGPRs and
      // restores them (v_accvgpr_read = a VALU write) right in front of the first S MFMA, which — being an asm — it does
      // not pad: the MFMA then reads stale registers (row block 0 only, data-dependent on timing).  Pin them here, two
      // wait states ahead.  tools/audit_mfma_hazard.py checks every MFMA of the generated assembly for this at build time.
      static_assert(PF == 2 || PF == 3, "pin list below");
      if constexpr (PF == 2) asm volatile("s_nop 1" : "+v"(kf[0]), "+v"(kf[1]));
      else asm volatile("s_nop 1" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]));
      // the softmax work of MFMA slot g, issued in this order behind the MFMA: scale/subtract (independent of everything
      // recent), then sum + pack of the elements exponentiated in the PREVIOUS slot, then this slot's exponentials.  A
      // transcendental result needs one wait state before a VALU instruction reads it, and hipcc counts an asm MFMA as
      // none: with the consumer as the first instruction behind the MFMA it pads every slot with an s_nop.
      auto soft_slot = [&](auto g_c) {
        constexpr int g = decltype(g_c)::value;
        if constexpr ((AB & X4AB_NOEXP) != 0) {
          static_for<0, NE>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, rb = X4_EL_RB(n), e = X4_EL_E(n);
            if constexpr (X4_SLOT_OF(n) == g && (e & 1)) {
              pw[rb][e >> 1] = __builtin_bit_cast(unsigned, scur[rb][e >> 4][e & 15]);
              asm volatile("" : "+v"(pw[rb][e >> 1]));
            }
          });
        } else {
          static_for<0, NE>([&](auto n_c) {
            constexpr int sc_n = X4_SLOT_OF(decltype(n_c)::value);
            if constexpr ((sc_n >= 2 ? sc_n - 2 : 0) == g) st_fma(n_c, scur, xs);
          });
          static_for<0, NE>([&](auto n_c) {
            if constexpr (X4_SLOT_OF(decltype(n_c)::value) == g) st_sum(n_c, xs, pw);
          });
          static_for<0, NE>([&](auto n_c) {
            if constexpr (X4_SLOT_OF(decltype(n_c)::value) - 1 == g) st_exp(n_c, xs);
          });
        }
      };
      // DMA piece n of this iteration (V(j+1) pieces 0..PPW-1, then K(j+3)) goes behind MFMA slot DMA0 + n * DMASTEP.  When
      // K(j+3) does not exist the piece is still issued, with an out-of-range offset (the descriptor's bounds check makes
      // it read zeros into a buffer nobody reads again): one v_cndmask instead of a branch in the tile body.
      const int k_tile_off = (j + 3) * k_tile_stride;
      auto dma_slot = [&](auto g_c) {
        constexpr int g = decltype(g_c)::value;
        if constexpr (!(AB & X4AB_NODMA) && g >= TFA_X4_DMA0 && (g - TFA_X4_DMA0) % TFA_X4_DMASTEP == 0 && (g - TFA_X4_DMA0) / TFA_X4_DMASTEP < 2 * PPW) {
          constexpr int n = (g - TFA_X4_DMA0) / TFA_X4_DMASTEP;
          if constexpr (n < PPW) dma_v1(j + 1, PAR ^ 1, n);
#if defined(TFA_X4_BRANCHY_K)
          else { if (issue_k) dma_k1(j + 3, kb0, n - PPW); }
#else
          else if constexpr (WIN) dma_k_win(j + 3, kb0, n - PPW, issue_k ? k_src[n - PPW] : (int)TFA_OOB);
          else lds_dma16_m0(k_rs, my_piece0 + kb0 + (n - PPW) * 1024, issue_k ? k_src[n - PPW] + k_tile_off : (int)TFA_OOB);
#endif
        }
      };
      static_assert(TFA_X4_DMA0 + (2 * PPW - 1) * TFA_X4_DMASTEP < N1 + N2, "a DMA piece falls behind the last MFMA slot");
      __builtin_amdgcn_sched_barrier(0);
      // part 1: S(j+1) = K(j+1) Q^T — fragment i feeds the MFMAs 2i (row block 0) and 2i+1 (row block 1)
      static_for<0, N1>([&](auto g_c) {
        constexpr int g = decltype(g_c)::value, i = g / RB, rb = g % RB;
        if constexpr (rb == 0) {                       // read-ahead: K fragments, then the first V fragments of part 2
          if constexpr (i + PF < NKF) kf[i + PF] = (AB & X4AB_NOKREAD) ? kf[i % PF] : k_frag(kb1, i + PF);
          else vf[i + PF - NKF] = v_frag(vbp, i + PF - NKF);
        }
        if constexpr ((AB & X4AB_NOQK) != 0) {
          if constexpr (i < 2) asm volatile("" : "+v"(snext[rb][i]));
        } else if constexpr ((i >> 1) == 0) {
          M::template qk0<X4_QBASE + 4 * (rb * DS + (i >> 1))>(snext[rb][i & 1], kf[i]);
        } else {
          M::template qk<X4_QBASE + 4 * (rb * DS + (i >> 1))>(snext[rb][i & 1], kf[i]);
        }
        dma_slot(g_c);
        soft_slot(g_c);
        __builtin_amdgcn_sched_barrier(0);
      });
      // tile j+1 is the wave's masked (diagonal / ragged) tile: S(j+1) is complete, mask it here, in a wave-uniform branch of
      // the fast path (the fence covers the MFMA -> VALU distance) instead of sending tile j through the burst path
      if (j + 1 >= fm) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          x4_fence_v(snext[rb][0]);
          x4_fence_v(snext[rb][1]);
          apply_mask(j + 1, rb, snext[rb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // part 2: O += P(j) V(j) — fragment i feeds the MFMAs RB*i .. RB*i + RB-1; row max of S(j+1): chain c = kt * RB + rb
      // finished at MFMA N1 - 2*RB + c of part 1 and is read from MFMA (N2 / (2*RB)) * c of part 2 on
      float mx[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) mx[rb] = -INFINITY;
      static_for<0, N2>([&](auto g_c) {
        constexpr int g = decltype(g_c)::value, i = g / RB, rb = g % RB;
        if constexpr (rb == 0) {
          if constexpr (i + PF < NVF) vf[i + PF] = (AB & X4AB_NOVREAD) ? vf[i % PF] : v_frag(vbp, i + PF);
#if defined(TFA_X4_BRANCHY_KPRE)
          else { if (j + 2 < nact) kpre[i + PF - NVF] = k_frag(kb2, i + PF - NVF); }
#else
          else kpre[i + PF - NVF] = k_frag(kb2, i + PF - NVF);   // K(j+2) landed before the previous barrier (read even when no later
#endif                                                           // iteration wants it: stale LDS bytes, never used — no branch)
        }
        if constexpr (!(AB & X4AB_NOPV)) M::template pv<(rb * DT + i % DT) * 16>(vf[i], p_frag(pw, rb, i / DT));
        else asm volatile("" ::"v"(vf[i]), "v"(pw[rb][(i / DT) * 4]), "v"(pw[rb][(i / DT) * 4 + 1]), "v"(pw[rb][(i / DT) * 4 + 2]), "v"(pw[rb][(i / DT) * 4 + 3]));
        dma_slot(std::integral_constant<int, N1 + g>{});
        soft_slot(std::integral_constant<int, N1 + g>{});
        if constexpr (!(AB & X4AB_NOMAX)) {
          static_for<0, 16 * RB>([&](auto q_c) {       // 16 pairs of S(j+1) values per row block -> one v_max3 each
            constexpr int q = decltype(q_c)::value;
            if constexpr (q * N2 / (16 * RB) == g) {
              constexpr int c = q >> 3, kt = c / RB, r2 = c % RB;
              mx[r2] = fmaxf(fmaxf(mx[r2], snext[r2][kt][2 * (q & 7)]), snext[r2][kt][2 * (q & 7) + 1]);
              asm volatile("" : "+v"(mx[r2]));
            }
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) mnext[rb] = (AB & X4AB_NOMAX) ? snext[rb][0][0] * 1e-30f : mx[rb];
      if (dbg & 16) big_fence();
      iter_end();
    };
    // ---- slow path: any tile (masked successor, last, re-base needed), burst-structured -------------------------------
    auto slow = [&](int j, f32x16 (&scur)[RB][2], const float (&mcur)[RB], f32x16 (&snext)[RB][2], float (&mnext)[RB]) {
      const int par = j & 1;
      ++n_slow;
      if (dbg & 2) big_fence();
      if (j + 3 < nt) dma_k(j + 3, kb0);
      if (j + 1 < nt) dma_v(j + 1, par ^ 1);
      const char* vbp = vl + par * TILE_BYTES;
      unsigned pw[RB][16];
      static_for<0, RB>([&](auto rb_c) {
        constexpr int rb = decltype(rb_c)::value;
        rescale_if_needed(rb_c, mcur[rb]);
        const float msc = mref[rb];
#pragma unroll
        for (int e = 0; e < 32; e += 2) {
          const int slot = e >> 3, t = slot >> 1, r = (slot & 1) * 8 + (e & 7);
          const float e0 = fast_exp2(fmaf(scur[rb][t][r], sc, -msc));
          const float e1 = fast_exp2(fmaf(scur[rb][t][r + 1], sc, -msc));
          l4[rb][e & 3] += e0;
          l4[rb][(e + 1) & 3] += e1;
          const t2 w = {(T)e0, (T)e1};
          pw[rb][e >> 1] = __builtin_bit_cast(unsigned, w);
        }
      });
      static_for<0, NVF>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        const X8 vfr = v_frag(vbp, i);
        X8 p0 = p_frag(pw, 0, i / DT), p1 = p_frag(pw, RB - 1, i / DT);
        asm volatile("s_nop 1" : "+v"(p0), "+v"(p1));    // VALU write -> MFMA operand: 2 wait states
        if (dbg & 4) big_fence();
        M::template pv<(0 * DT + i % DT) * 16>(vfr, p0);
        if constexpr (RB == 2) M::template pv<(1 * DT + i % DT) * 16>(vfr, p1);
      });
      if (j + 1 < nact) qk_burst(kb1, j + 1, snext, mnext);
      if (!(dbg & 32) && j + 2 < nact) load_kpre(kb2);
      iter_end();
    };

    // S(j) lives in sA for even j and in sB for odd j on both paths, so the paths alternate freely without copies.
    // Tile j takes the fast path when tile j+1 exists for this wave and no row max of tile j has outgrown its reference;
    // the wave's last tile and re-bases take the slow path.
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // ---- the hand-scheduled steady state (round 5; tfa_fwd_x4_asm_loop.inc, generated by tools/gen_x4_asm_loop.py): the 256-wide form (one 32-row
    // block per wave; five to eight valid column blocks: head dims 136..256, one text each).  Six tile bodies (K ring of three x V pair): entered at a tile j with j % 6 == 0 — the pass's first tile, or
    // wherever the compiler-scheduled bodies realign after a re-base —, it runs while the next tile exists for the wave, is unmasked, K(j+3) exists and no row
    // outgrew its reference, and returns the first tile it did not process: S(j) in sA / sB by parity, its half-wave row maximum in mA / mB, the first
    // two fragments of K(j+1) in kpre.  The ring offsets are then those of tile j (K(t) lives in ring buffer t % 3).
    // (its LDS-DMA requests carry the tile's byte offset as their SCALAR offset, which the bounds check does not see: only tiles wholly inside the key sequence)
    const int nt_full = (p.Nk / BN) < nt ? (p.Nk / BN) : nt;
    constexpr bool ASMX4 = D == 256 && RB == 1 && DVB >= 5 && DVB <= 8 && AB == 0 && PF == 2 && PPW == 8 && !WIN && TFA_X4_USE_ASMLOOP;
    auto asm_loop = [&](int& j) {
      if constexpr (ASMX4) {
        const int lim = fm < nact ? fm : nact;
        const int jend = (lim - 1 < nt_full - 3) ? lim - 1 : nt_full - 3;   // tiles j < jend take the loop
        int koff = (j + 3) * k_tile_stride, voff = (j + 1) * v_tile_stride;
        const unsigned vaddr = lds_base + NKB * TILE_BYTES + (unsigned)v_rd_base;
        const unsigned ldsw = __builtin_amdgcn_readfirstlane(my_piece0);
        float thr;
        u32x4 f2, f3, f4, f5, f6, f7;
        typedef __attribute__((ext_vector_type(16))) unsigned int u32x16;
#if TFA_X4_ASM_NBUF == 8
#define TFA_X4_ASM_FRAGS [f2] "=&v"(f2), [f3] "=&v"(f3), [f4] "=&v"(f4), [f5] "=&v"(f5), [f6] "=&v"(f6), [f7] "=&v"(f7)
#else
#define TFA_X4_ASM_FRAGS [f2] "=&v"(f2), [f3] "=&v"(f3)
#endif
        u32x16 ka, kb;
#define TFA_X4_ASM_STMT(TEXT)                                                                                                                              \
        asm volatile(TEXT                                                                                                                                  \
                     : [sa0] "+v"(sA[0][0]), [sa1] "+v"(sA[0][1]), [sb0] "+v"(sB[0][0]), [sb1] "+v"(sB[0][1]),                                            \
                       [l0] "+v"(l4[0][0]), [l1] "+v"(l4[0][1]), [l2] "+v"(l4[0][2]), [l3] "+v"(l4[0][3]), [ma] "+v"(mA[0]), [mb] "+v"(mB[0]), [j] "+s"(j), \
                       [koff] "+s"(koff), [voff] "+s"(voff), [f0] "+v"(kpre[0]), [f1] "+v"(kpre[1]),                                                      \
                       TFA_X4_ASM_FRAGS, [ka] "=&v"(ka), [kb] "=&v"(kb), [thr] "=&v"(thr)                                                                  \
                     : [mref] "v"(mref[0]), [kaddr] "v"(k_rd_addr), [va] "v"(vaddr),                                                                       \
                       [ks0] "v"(k_src[0]), [ks1] "v"(k_src[1]), [ks2] "v"(k_src[2]), [ks3] "v"(k_src[3]), [ks4] "v"(k_src[4]), [ks5] "v"(k_src[5]),      \
                       [ks6] "v"(k_src[6]), [ks7] "v"(k_src[7]), [vs0] "v"(v_src[0]), [vs1] "v"(v_src[1]), [vs2] "v"(v_src[2]), [vs3] "v"(v_src[3]),      \
                       [vs4] "v"(v_src[4]), [vs5] "v"(v_src[5]), [vs6] "v"(v_src[6]), [vs7] "v"(v_src[7]),                                                \
                       [sc] "s"(sc), [krs] "s"(k_rs), [vrs] "s"(v_rs), [ldsw] "s"(ldsw), [kstr] "s"(k_tile_stride), [vstr] "s"(v_tile_stride), [jend] "s"(jend) \
                     : TFA_X4_ALLCLOB, "m0", "vcc", "scc", "memory")
        constexpr bool BF = std::is_same<T, __bf16>::value;      // one text per (dtype, count of valid 32-column blocks)
#if defined(TFA_X4_MF_ARM)     // timing arm (round 6; needs the file generated with TFA_GEN_X4_MF=1): the max-free texts under the lazy rule's C++ — same bits while no row
                               // outgrows its reference.  +0.7 .. +1.1 % (profiles/r06_x4_maxfree_arm.txt): not built into a product rule for this kernel
        if constexpr (DVB == 8 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V8_MF); }
        else if constexpr (DVB == 7 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V7_MF); }
        else if constexpr (DVB == 6 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V6_MF); }
        else if constexpr (DVB == 5 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V5_MF); }
        else
#endif
        if constexpr (DVB == 8 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V8); }
        else if constexpr (DVB == 8) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V8_F16); }
        else if constexpr (DVB == 7 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V7); }
        else if constexpr (DVB == 7) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V7_F16); }
        else if constexpr (DVB == 6 && BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V6); }
        else if constexpr (DVB == 6) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V6_F16); }
        else if constexpr (BF) { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V5); }
        else { TFA_X4_ASM_STMT(TFA_X4_ASM_LOOP_V5_F16); }
#undef TFA_X4_ASM_STMT
#undef TFA_X4_ASM_FRAGS
        (void)f4; (void)f5; (void)f6; (void)f7;
        kb0 = (unsigned)(j % 3) * TILE_BYTES;
        kb1 = (unsigned)((j + 1) % 3) * TILE_BYTES;
        kb2 = (unsigned)((j + 2) % 3) * TILE_BYTES;
      }
    };
#pragma nounroll
    for (int j = 0; j < nact; j += 2) {
      bool ring_set = false;
      if (ASMX4 && !(dbg & 1) && (j % 6) == 0 && j + 1 < nact && j + 1 < fm && j + 3 < nt_full && !trigger(mA)) {
        int jj = j;
        asm_loop(jj);                                  // tiles j .. jj-1 done, jj > j
        ring_set = true;
        if (jj & 1) j = jj - 1;                        // on to the odd step below, for tile jj
        else { j = jj - 2; continue; }                 // back to the even step, for tile jj
      }
      else if (!(dbg & 1) && j + 1 < nact && !trigger(mA)) fused(C0{}, j, sA, sB, mB);
      else { if (p.trace && j + 1 < nact) ++n_trig; slow(j, sA, mA, sB, mB); }
      if (!ring_set) rotate();
      if (j + 1 >= nact) break;
      if (!(dbg & 1) && j + 2 < nact && !trigger(mB)) fused(C1{}, j + 1, sB, sA, mA);
      else { if (p.trace && j + 2 < nact) ++n_trig; slow(j + 1, sB, mB, sA, mA); }
      rotate();
    }
#pragma nounroll
    for (int j = nact; j < nt; ++j) {                  // tiles of the block this wave does not touch: its DMA pieces and barriers
      if (j + 3 < nt) dma_k(j + 3, kb0);
      if (j + 1 < nt) dma_v(j + 1, (j & 1) ^ 1);
      iter_end();
      rotate();
    }
    if (p.trace && pass == tr_pass) t_loop = __builtin_amdgcn_s_memtime();

    // ---- epilogue ---------------------------------------------------------------------------------------------
    if (dbg & 64) big_fence();
    x4_o_fence();
    static_for<0, RB>([&](auto rb_c) {
      constexpr int rb = decltype(rb_c)::value;
      const int my_row = wave_row0 + rb * 32 + qi;
      const float l_tot = pair_sum((l4[rb][0] + l4[rb][1]) + (l4[rb][2] + l4[rb][3]));
      const bool empty = !(l_tot > 0.f);
      const float inv = empty ? 1.f : 1.f / l_tot;
      if (p.lse != nullptr && hi == 0 && my_row < p.Nq) {
        const float lse = empty ? INFINITY : (mref[rb] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
        p.lse[(long long)bh * p.Nq + my_row] = lse;
      }
      if constexpr (F32OUT) {
        float* obase = reinterpret_cast<float*>(p.o) + b * p.os_b + h * p.os_h;
        auto o_rs = slice_rsrc(obase, p.o_bytes, (unsigned long long)(unsigned)q0 * (unsigned long long)p.os_n * 4);
        const int ooff = (my_row - (WIN ? q0 : 0)) * (int)p.os_n * 4 + hi * 16;
        static_for<0, DT * 4>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value, R = (rb * DT + c / 4) * 16 + (c % 4) * 4;
          f32x4 v4 = {x4_o_read<R>() * inv, x4_o_read<R + 1>() * inv, x4_o_read<R + 2>() * inv, x4_o_read<R + 3>() * inv};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), o_rs, (c / 4) * 32 + (c % 4) * 8 + hi * 4 < p.dv ? ooff + ((c / 4) * 32 + (c % 4) * 8) * 4 : (int)TFA_OOB, 0, 0);
        });
      } else if constexpr (EPI) {
        // A lane holds 4-element pieces of ONE row in 16 register groups: stored directly that is 16 eight-byte stores per
        // lane, 32 different rows per instruction.  Instead the wave transposes each 32 x D block through its own slice
        // of the epilogue region (16-byte chunk index XOR row, as for K) and writes whole rows: 1 KiB contiguous per store.
        T* obase = reinterpret_cast<T*>(p.o) + b * p.os_b + h * p.os_h;
        auto o_rs = slice_rsrc(obase, p.o_bytes, (unsigned long long)(unsigned)q0 * (unsigned long long)p.os_n * 2);
        typedef __attribute__((ext_vector_type(4))) T t4;
        int qix = qi, lanex = lane;                  // (through an empty asm: none of the addresses below may be hoisted out of the pass loop)
        asm volatile("" : "+v"(qix), "+v"(lanex));
        char* const ow = smem + (EPI_IN ? 0 : (NKB + 2) * TILE_BYTES) + (wave * RB + rb) * (32 * D * 2);
        constexpr int CH = D / 8;                    // 16-byte chunks per row
        const int osw = (CH >= 16) ? (qix & 15) : (qix & 7);
        static_for<0, DT * 4>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value, R = (rb * DT + c / 4) * 16 + (c % 4) * 4;
          t4 v4 = {(T)(x4_o_read<R>() * inv), (T)(x4_o_read<R + 1>() * inv), (T)(x4_o_read<R + 2>() * inv), (T)(x4_o_read<R + 3>() * inv)};
          *reinterpret_cast<u32x2*>(ow + qix * (D * 2) + ((c ^ osw) << 4) + (lanex >> 5) * 8) = __builtin_bit_cast(u32x2, v4);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no barrier needed
        constexpr int RPI = 64 / CH;                 // rows per store instruction (4 at D=128, 8 at D=64)
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
          const int r = i * RPI + lanex / CH, cpos = lanex % CH;
          const int c = cpos ^ ((CH >= 16) ? (r & 15) : (r & 7));
          u32x4 v = *reinterpret_cast<const u32x4*>(ow + r * (D * 2) + (cpos << 4));
          __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, c * 8 < p.dv ? (wave_row0 - (WIN ? q0 : 0) + rb * 32 + r) * (int)p.os_n * 2 + (c << 4) : (int)TFA_OOB, 0, 2);   // nt: see tfa_fwd_kernel_il.h
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is rewritten by this wave's next epilogue only
      } else {
        T* obase = reinterpret_cast<T*>(p.o) + b * p.os_b + h * p.os_h;
        auto o_rs = slice_rsrc(obase, p.o_bytes, (unsigned long long)(unsigned)q0 * (unsigned long long)p.os_n * 2);
        const int ooff = (my_row - (WIN ? q0 : 0)) * (int)p.os_n * 2 + hi * 8;
        typedef __attribute__((ext_vector_type(4))) T t4;
        static_for<0, DT * 4>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value, R = (rb * DT + c / 4) * 16 + (c % 4) * 4;
          t4 v4 = {(T)(x4_o_read<R>() * inv), (T)(x4_o_read<R + 1>() * inv), (T)(x4_o_read<R + 2>() * inv), (T)(x4_o_read<R + 3>() * inv)};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v4), o_rs, (c / 4) * 32 + (c % 4) * 8 + hi * 4 < p.dv ? ooff + ((c / 4) * 32 + (c % 4) * 8) * 2 : (int)TFA_OOB, 0, 0);
        });
      }
    });
    // (the next pass's first DMA pieces land in the K/V buffers: every wave is past its last tile's reads — the last
    //  iteration ended with a barrier — and the epilogue slices are private; in place they lie IN those buffers: every wave's rows must be out)
    if (EPI_IN && !F32OUT && pass + 1 < npass) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  if (p.trace) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
      t[0] = t_start; t[1] = t_pro; t[2] = t_loop; t[3] = t_end;
      t[4] = (unsigned long long)nt_total | ((unsigned long long)n_slow << 32) | ((unsigned long long)n_trig << 48);   // wave 0: slow-path tiles, of which re-base triggers
      t[5] = (unsigned long long)__builtin_amdgcn_s_getreg(63508) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32);   // XCC_ID | HW_ID << 32
      t[6] = __builtin_amdgcn_s_memrealtime() - rt_start;   // 100 MHz ticks over the same span as t[3] - t[0] shader cycles
      t[7] = ((unsigned long long)bh << 32) | (unsigned)wi;
    }
  }
}

#undef X4_SLOT_OF
#undef X4_EL_RB
#undef X4_EL_E

}  // namespace tfa
