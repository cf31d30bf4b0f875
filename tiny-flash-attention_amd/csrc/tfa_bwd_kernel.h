// tfa_bwd_kernel.h — FlashAttention-2 backward for gfx950, first correct version.
//
// The reference has no backward; it only saves the LSE for one ("LogSumExp save for backward",
// flash_attention_cutlass/csrc/flash_attention.cu:353-354,614-623; tiny_flash_attn_triton.py:27-29).  The math is
// the standard one:  P = exp(S*scale - LSE),  dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta) * scale with
// delta_i = sum_d dO_id O_id,  dQ = dS K,  dK = dS^T Q.
//
// ONE kernel template, three launches (dQ, dK, dV), each a role assignment of the forward kernel's two GEMM forms, so
// every layout below is the forward's, already verified on hardware:
//   * a wave owns 32 RESIDENT rows (one per lane: query rows for dQ, key rows for dK/dV) whose 16-bit fragments stay in
//     registers as MFMA B operands; the other sequence is STREAMED in 64-row tiles through LDS by LDS-DMA;
//   * "GEMM-I"  X^T[tile row, resident row] = Tile . Resident^T  (tile image in the K layout: row-major, XOR swizzle)
//       gives S (and dP) with the resident row in the lane and 16 tile rows per accumulator;
//   * "GEMM-II" Acc^T[d, resident row] += Tile^T . Y  (tile image in the V layout: ds_read_b64_tr_b16) with Y = P or dS
//       taken straight from the GEMM-I accumulator registers (16-bit pack, no data movement).
//   dQ: resident Q, dO;  tiles K (K layout), V (K layout), K (V layout);  row statistics (LSE, delta) per lane.
//   dK: resident K, V;   tiles Q (K layout), dO (K layout), Q (V layout); statistics per TILE row (loaded per tile).
//   dV: resident K;      tiles Q (K layout), dO (V layout).
// No atomics, no transposes through LDS, deterministic; the price is that S is recomputed three times and dP twice
// (8 GEMM units instead of 5).  Burst-structured like tfa_fwd_kernel_dma.h (two LDS stages, one barrier per tile);
// the issue-interleaving of tfa_fwd_kernel_il.h is the next step for this kernel.
#pragma once
#include "tfa_fwd_kernel_dma.h"
#include "tfa_acc_regs.h"
#if defined(TFA_BWD_DQ_ASM_INC)
#include TFA_BWD_DQ_ASM_INC       // an A/B arm's text (tools/gen_bwd_dq_asm_loop.py with TFA_GEN_DQ_* set), never the product build
#else
#include "tfa_bwd_dq_asm_loop.inc"
#endif

#if !defined(TFA_BWD_DQ_USE_ASM)
#define TFA_BWD_DQ_USE_ASM 1     // 0: the compiler-scheduled tile body everywhere (the A/B arm of the hand-scheduled dQ tiles, tools/gen_bwd_dq_asm_loop.py)
#endif
// The hand-scheduled unmasked tiles of the dQ launch: ONE statement (one register assignment), every operand the compiler's choice.
#define TFA_BWD_DQ_ASM_STMT(TEXT) \
  asm volatile(TEXT \
  : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [acc3] "+v"(acc[3]), [u] "+s"(ua), [koff] "+s"(koff), [voff] "+s"(voff), [lim] "+v"(a_lim), [ts] "=&s"(a_ts), [msk] "=&s"(a_msk), [ninf] "=&v"(a_ninf), \
  [s0] "=&v"(as0), [p0] "=&v"(ap0), [s1] "=&v"(as1), [p1] "=&v"(ap1), [f0] "=&v"(af0), [f1] "=&v"(af1), [f2] "=&v"(af2), [f3] "=&v"(af3), \
  [ka] "=&v"(aka), [ka5] "=&v"(aka5), [ka6] "=&v"(aka6), [ka7] "=&v"(aka7) \
  : [q0] "v"(r1f[0]), [q1] "v"(r1f[1]), [q2] "v"(r1f[2]), [q3] "v"(r1f[3]), [q4] "v"(r1f[4]), [q5] "v"(r1f[5]), [q6] "v"(r1f[6]), [q7] "v"(r1f[7]), \
  [d0] "v"(r2f[0]), [d1] "v"(r2f[1]), [d2] "v"(r2f[2]), [d3] "v"(r2f[3]), [d4] "v"(r2f[4]), [d5] "v"(r2f[5]), [d6] "v"(r2f[6]), [d7] "v"(r2f[7]), \
  [kaddr] "v"(a_kaddr), [vat] "v"(a_vat), [ks0] "v"(src[0][0]), [ks1] "v"(src[0][1]), [vs0] "v"(src[1][0]), [vs1] "v"(src[1][1]), \
  [ts0] "v"(src[2][0]), [ts1] "v"(src[2][1]), [l2] "v"(lse2_lane), [dinit] "v"(dinit), \
  [sc] "s"(a_sc), [krs] "s"(rs_fixed[0]), [vrs] "s"(rs_fixed[1]), [ldsw] "s"(a_ldsw), [kstr] "s"(a_kstr), [vstr] "s"(a_vstr), [uend] "s"(a_uend), [mend] "s"(a_mend), [nu] "s"(a_nu) \
  : "m0", "vcc", "scc", "memory")

namespace tfa {

// c + sum of the eight products a[e] * b[e], fp32 accumulation (v_dot2c_f32_bf16 / v_dot2c_f32_f16): delta inside the dQ launch
template <typename T> struct Dot8;
template <> struct Dot8<__bf16> {
  static __device__ __forceinline__ float f(bf16x8 a, bf16x8 b, float c) {
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c, false);
    return c;
  }
};
template <> struct Dot8<_Float16> {
  static __device__ __forceinline__ float f(f16x8 a, f16x8 b, float c) {
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c, false);
    return c;
  }
};

struct BTensor {
  const void* p;
  long long s_b, s_h, s_n;   // strides in elements; unit stride along D
  unsigned bytes;            // extent of one (b,h) slice in bytes (buffer descriptor range; clamped to 2 GiB - 1 when the slice is larger)
  unsigned long long full;   // the slice's true extent: the BIG instantiations address it through windows (rsrc_at)
};

struct BArgs {
  BTensor q, k, v, dout;
  BTensor out;               // the forward's O: read by the dQ launch when it computes delta itself (fuse_delta)
  float* delta_w;            // ... and writes it here (= delta) for the launches behind it
  int fuse_delta;            // dQ launch: delta[b,h,i] = sum_d dO.O from its resident dO rows and the O rows, instead of a launch of its own (51 us at config 3)
  void* grad;                // output of this launch: dQ, dK or dV
  long long gs_b, gs_h, gs_n;
  unsigned g_bytes;
  const float* lse;          // (B,H,Nq) natural-log LSE from the forward
  const float* delta;        // (B,H,Nq) rowsum(dO o O)
  int B, H, Hk, Nq, Nk;
  int dv;                    // valid head dim (multiple of 8, <= the kernel's width D): 16-byte chunks beyond it are read as zeros
                             // (their offsets point out of the descriptor's range, TFA_OOB) and never stored
  int nrb;                   // resident blocks per (b, resident head): 256 rows (bwd_kernel) or 128 keys (bwd_kv_kernel)
  float scale, scale_log2;
  void* grad2;               // bwd_kv_kernel (tfa_bwd_kv_kernel.h): `grad` = dK, `grad2` = dV
  long long g2s_b, g2s_h, g2s_n;
  unsigned g2_bytes;
  unsigned long long g_full, g2_full;   // true extents of the gradient slices (BIG)
  int big;                   // some slice reaches 2 GiB: the host launches the BIG instantiations
  void* ws;                  // optional dS workspace (tfa_bwd_params::workspace): dS^T[b][query head][ws_nk key rows][ws_nq queries], 16 bit
  int ws_nk, ws_nq;          // padded extents: Nk rounded up to 128, Nq rounded up to 256
#if defined(TFA_BWD_TRACE)
  void* tr;                  // debug build (tools/trace_bwd_kv.py): 4 x uint64 per wave of the fused dK/dV launch
#endif
};

enum { BWD_DQ = 0, BWD_DK = 1, BWD_DV = 2 };

// Unified tile image (UNI, off by default): ONE row-major image per streamed tensor serves both GEMM forms, so a tile costs
// two LDS images instead of three.  Correct and conflict-free (SQ_LDS_BANK_CONFLICT = 0) but 4 % SLOWER than the
// three-image version on non-causal shapes (same-process A/B, tools/ab_libs.py --bwd), neutral on causal ones.  The 16-byte chunk index of row r is XORed with u_swz(r): bijective over 16 consecutive
// rows (ds_read_b128 of GEMM-I: 16 rows x one chunk hit 16 different slots), and for the four consecutive rows a
// ds_read_b64_tr_b16 group addresses, the 64-byte spans (4 chunks) land in different aligned groups of 4 chunks
// (D=128), resp. in the other half of the 256-byte bank row (D=64).
template <int D> static __device__ __forceinline__ int u_swz(int row) {
  // (D = 256: 32 chunks per 512-byte row; the XOR stays inside a 16-chunk half, and a row is two whole 256-byte bank rows, so the
  //  bank of a chunk depends on (chunk ^ swizzle) alone: the D = 128 pattern serves)
  return D >= 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
}

// NW = 8 (two waves per SIMD; head dims up to 128) or 4 (ONE wave per SIMD with the whole 512-entry register file: head dims up to 256 —
// 128 resident-fragment + 128 accumulator registers in the dK launch —, unified images; hipcc places the accumulators in AGPRs)
// BIG: (b,h) slices of 2 GiB and more (long (B,N,H,D) tensors): every descriptor is a WINDOW (rsrc_at) — one per streamed tile, one
// over the workgroup's resident rows, one over its gradient rows — and the 32-bit offsets are window-relative.  Its own instantiation
// (a fresh descriptor per tile costs scalar work and wait states), launched only when a slice needs it; dQ mode only at head dims up to 128 (the fused
// dK/dV launch has its own), all three modes of the 256-wide kernel (round 5).
// DVB: 32-wide column blocks that can hold valid head-dim columns — ceil(head dim / 32): 5..8 in the 256-wide kernels, 3 (head dims
// up to 96) in the 128-wide dQ kernel, 1 (up to 32) in the 64-wide one.  The LDS images
// and every address stay 256 wide (the columns beyond the head dim are the descriptors' zeros); the GEMM loops run over the valid
// blocks only: at D = 160 / 192 / 224 that is 5/8, 6/8, 7/8 of the matrix work, fragment registers and LDS reads.
template <typename T, int D, int MODE, bool CAUSAL, bool F32OUT, bool UNI = false, int NW = 8, bool BIG = false, int DVB = D / 32>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 1 : 2) void bwd_kernel(const BArgs p) {
  static_assert(!BIG || MODE == BWD_DQ || D > 128, "BIG: the dQ launch; at head dims above 128 (no fused dK/dV launch) all three");
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int BM = NW * 32;                      // resident rows per workgroup
  constexpr int BN = 64;                           // streamed rows per tile
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NW;
  static_assert(DVB >= 1 && DVB <= D / 32, "valid 32-column blocks of a D-wide kernel");
  constexpr int DS = 2 * DVB;                      // k-steps of 16 columns (GEMM-I), resident fragments per tensor
  constexpr int DT = DVB;                          // 32-column tiles of the gradient (GEMM-II)
  constexpr int DT_L = D / 32;                     // ... as the LDS layouts count them
  constexpr bool KEYS_RES = MODE != BWD_DQ;        // resident rows are keys
  constexpr bool NEED_DP = MODE != BWD_DV;
  constexpr int NIMG = (UNI || MODE == BWD_DV) ? 2 : 3;   // LDS images per tile
  constexpr int IMG_TR = UNI ? (MODE == BWD_DV ? 1 : 0) : NIMG - 1;   // the image GEMM-II reads (UNI: dQ: K, dK: Q, dV: dO)
  static_assert(PPW >= 1 && PPW * NW == PIECES, "");
  // the hand-scheduled unmasked tiles (tfa_bwd_dq_asm_loop.inc): the dQ launch of the 128-wide kernel.  Its LDS map puts the four row-major images first — every
  // ds_read_b128 offset of the statement fits the instruction's 16-bit immediate — and the two transposed K images behind them
  constexpr bool ASMDQ = TFA_BWD_DQ_USE_ASM != 0 && MODE == BWD_DQ && D == 128 && NW == 8 && !UNI && !BIG && DVB == 4;
  auto img_off = [](int stage, int img) -> int {
    if (ASMDQ) return img < 2 ? (2 * stage + img) * TILE_BYTES : (4 + stage) * TILE_BYTES;
    return (stage * NIMG + img) * TILE_BYTES;
  };

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The two waves of a SIMD run the burst-structured tile body (GEMM-I, element-wise, GEMM-II) in step behind the tile barrier: matrix bursts collide,
  // element-wise bursts leave the pipe idle.  A static issue priority for the younger wave (role 1 in the fused launch) lets it win the first
  // burst and pulls the two out of step: +0.3 .. +0.8 percent at D = 128 over five shapes, -2.6 percent at D = 64 (profiles/r04b_bwd_priority_ab.txt)
  if (D == 128 && NW == 8 && (wave >= NW / 2)) asm volatile("s_setprio 2" ::: "memory");
  const int qi = lane & 31;
  const int hi = lane >> 5;

  const int G = p.H / p.Hk;
  const int Hres = KEYS_RES ? p.Hk : p.H;
  // Workgroups are dispatched in order and round-robin over the 8 XCDs: keep a (b, head) on one XCD (its streamed
  // tensors are re-read by every resident block: L2 reuse) and hand out the blocks with the MOST tiles first
  // (causal dQ: the last query block; causal dK/dV: the first key block), so the tail of the launch is short.
  int rb, bhr;
  {
    const int nbh = p.B * Hres, id = blockIdx.x;
    if ((nbh & 7) == 0) {
      const int x = id & 7, q8 = id >> 3;
      bhr = x + 8 * (q8 / p.nrb);
      rb = q8 % p.nrb;
    } else {
      bhr = id / p.nrb;
      rb = id % p.nrb;
    }
    if (CAUSAL && !KEYS_RES) rb = p.nrb - 1 - rb;
  }
  const int b = bhr / Hres;
  const int hr = bhr - b * Hres;
  const int shift = p.Nk - p.Nq;
  const int r0 = rb * BM;
  const int wave_row0 = r0 + wave * 32;
  const int my_row = wave_row0 + qi;
  const int NG = KEYS_RES ? G : 1;                 // streamed heads per resident head

  // ---- streamed tile range of this block -----------------------------------------------------------------------
  int t_begin = 0, t_end;
  if (!KEYS_RES) {
    int kv_end = p.Nk;
    if (CAUSAL) {
      const int lim = r0 + BM + shift;
      kv_end = lim < kv_end ? lim : kv_end;
    }
    t_end = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
  } else {
    t_end = (p.Nq + BN - 1) / BN;
    if (CAUSAL) {
      const int qmin = r0 - shift;                 // first query row that sees key r0
      t_begin = qmin > 0 ? qmin / BN : 0;
      if (t_begin > t_end) t_begin = t_end;
    }
  }
  const int ntl = t_end - t_begin;                 // tiles per streamed head
  const int nu = ntl * NG;                         // tiles of the whole flat sequence (head-major)

  // ---- per-lane DMA source offsets (bytes) of the images, relative to the streamed (b,h) slice -------------------
  // image tensors: dQ: K, V, K   dK: Q, dO, Q   dV: Q, dO
  auto img_tensor = [&](int img) -> const BTensor& {
    if (MODE == BWD_DQ) return img == 1 ? p.v : p.k;
    return img == 1 ? p.dout : p.q;
  };
  int src[NIMG][PPW];
  int tile_stride[NIMG];
#pragma unroll
  for (int img = 0; img < NIMG; ++img) {
    const int sn = (int)img_tensor(img).s_n;
    tile_stride[img] = BN * sn * 2;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave * PPW + i;
      if (UNI) {                                     // unified row-major image
        const int row = pc * (1024 / (D * 2)) + lane / CPR;
        const int cpos = lane % CPR;
        const int ch = cpos ^ u_swz<D>(row);
        src[img][i] = ch * 8 < p.dv ? row * sn * 2 + (ch << 4) : (int)TFA_OOB;
      } else if (img != IMG_TR) {                    // K layout
        const int row = pc * (1024 / (D * 2)) + lane / CPR;
        const int cpos = lane % CPR;
        const int ch = cpos ^ k_swz<D>(row);
        src[img][i] = ch * 8 < p.dv ? row * sn * 2 + (ch << 4) : (int)TFA_OOB;
      } else {                                       // V layout
        const int o = pc * 1024 + lane * 16;
        const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
        const int dt = sub % DT_L, sh = sub / DT_L;
        const int row = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
        src[img][i] = (dt * 4 + pcs) * 8 < p.dv ? row * sn * 2 + ((dt * 4 + pcs) << 4) : (int)TFA_OOB;
      }
    }
  }
  // flat tile u -> (streamed head, tile index).  dQ (one streamed head per workgroup): the descriptors are built once; the
  // key-resident modes rebuild them per call (GQA walks G query heads)
  __amdgpu_buffer_rsrc_t rs_fixed[NIMG];
  if (!KEYS_RES) {
#pragma unroll
    for (int img = 0; img < NIMG; ++img) {
      const BTensor& x = img_tensor(img);
      const T* base = reinterpret_cast<const T*>(x.p) + b * x.s_b + (hr / G) * x.s_h;
      rs_fixed[img] = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, x.bytes, 0x00020000);
    }
  }
  auto dma_issue = [&](int u, int stage) {
    const int g = KEYS_RES ? u / ntl : 0;
    const int jt = t_begin + (u - g * ntl);
    const int hs = KEYS_RES ? (hr * G + g) : (hr / G);
#pragma unroll
    for (int img = 0; img < NIMG; ++img) {
      const BTensor& x = img_tensor(img);
      const T* base = reinterpret_cast<const T*>(x.p) + b * x.s_b + hs * x.s_h;
      if constexpr (BIG) {
        const auto rsw = rsrc_at(base, x.full, (unsigned long long)jt * (unsigned)tile_stride[img]);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
          lds_dma16_m0_fresh(rsw, lds_base + img_off(stage, img) + (wave * PPW + i) * 1024, src[img][i]);
        continue;
      }
      auto rs = KEYS_RES ? __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, x.bytes, 0x00020000) : rs_fixed[img];
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        lds_dma16_m0(rs, lds_base + img_off(stage, img) + (wave * PPW + i) * 1024, src[img][i] + jt * tile_stride[img]);
    }
    if (KEYS_RES && wave < 2) {
      // the tile's row statistics (LSE, delta: 64 rows = one dword per lane) travel with it — loaded from global memory by every lane
      // they queue behind the next tile's DMA pieces in the in-order vmcnt (tfa_bwd_kv_kernel.h)
      const long long so = (long long)(b * p.H + hs) * p.Nq;
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)((wave ? p.delta : p.lse) + so), 0, (unsigned)p.Nq * 4u, 0x00020000);
      lds_dma4_m0(rs, lds_base + 2 * NIMG * TILE_BYTES + stage * 512 + wave * 256, (jt * BN + lane) * 4);
    }
  };

  // ---- resident fragments and per-lane statistics -----------------------------------------------------------------
  X8 r1f[DS], r2f[NEED_DP ? DS : 1];
  {
    const BTensor& x1 = KEYS_RES ? p.k : p.q;
    const T* b1 = reinterpret_cast<const T*>(x1.p) + b * x1.s_b + hr * x1.s_h;
    auto rs1 = BIG ? rsrc_at(b1, x1.full, (unsigned long long)r0 * (unsigned long long)x1.s_n * 2ull) : __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, x1.bytes, 0x00020000);
    const int off1 = (my_row - (BIG ? r0 : 0)) * (int)x1.s_n * 2 + hi * 16;
#pragma unroll
    for (int s = 0; s < DS; ++s) r1f[s] = __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(rs1, (2 * s + hi) * 8 < p.dv ? off1 + s * 32 : (int)TFA_OOB, 0, 0));
    if (NEED_DP) {
      const BTensor& x2 = KEYS_RES ? p.v : p.dout;
      const T* b2 = reinterpret_cast<const T*>(x2.p) + b * x2.s_b + hr * x2.s_h;
      auto rs2 = BIG ? rsrc_at(b2, x2.full, (unsigned long long)r0 * (unsigned long long)x2.s_n * 2ull) : __builtin_amdgcn_make_buffer_rsrc((void*)b2, 0, x2.bytes, 0x00020000);
      const int off2 = (my_row - (BIG ? r0 : 0)) * (int)x2.s_n * 2 + hi * 16;
#pragma unroll
      for (int s = 0; s < DS; ++s) r2f[s] = __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(rs2, (2 * s + hi) * 8 < p.dv ? off2 + s * 32 : (int)TFA_OOB, 0, 0));
    }
  }
  // dQ, fuse_delta: this wave's O rows, for delta = rowsum(dO o O) (the dO rows are resident anyway): one launch and one pass over dO less
  const bool fuse = MODE == BWD_DQ && p.fuse_delta != 0;
  X8 of[MODE == BWD_DQ ? DS : 1];
  if (MODE == BWD_DQ && fuse) {
    const T* b3 = reinterpret_cast<const T*>(p.out.p) + b * p.out.s_b + hr * p.out.s_h;
    auto rs3 = BIG ? rsrc_at(b3, p.out.full, (unsigned long long)r0 * (unsigned long long)p.out.s_n * 2ull) : __builtin_amdgcn_make_buffer_rsrc((void*)b3, 0, p.out.bytes, 0x00020000);
    const int off3 = (my_row - (BIG ? r0 : 0)) * (int)p.out.s_n * 2 + hi * 16;
#pragma unroll
    for (int s = 0; s < DS; ++s) of[s] = __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(rs3, (2 * s + hi) * 8 < p.dv ? off3 + s * 32 : (int)TFA_OOB, 0, 0));
  }
  float lse2_lane = 0.f, delta_lane = 0.f;           // dQ: statistics of the lane's own query row
  const long long stat_i = (long long)(b * p.H + hr) * p.Nq + my_row;
  if (!KEYS_RES && my_row < p.Nq) {
    lse2_lane = p.lse[stat_i] * 1.4426950408889634f;
    if (!fuse) delta_lane = p.delta[stat_i];
  }

  constexpr bool OWN_ACC = NW == 4;                  // head dims above 128: the accumulators are the hand-owned a[0:127] (tfa_acc_regs.h)
  f32x16 acc[OWN_ACC ? 1 : DT];
  if (OWN_ACC) g_zero();
  else {
#pragma unroll
    for (int d = 0; d < (OWN_ACC ? 1 : DT); ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  }

  const int k_rd_base = qi * (D * 2);
  const int k_rd_swz = UNI ? u_swz<D>(qi) : k_swz<D>(qi);
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT_L << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  // UNI transpose reads: this lane addresses 4 consecutive d (8 bytes) of tile row 16*sl + tr_row (first read) and
  // 16*sl + tr_row + 8 (second read); chunk = 4*dtile + tr_clo, XOR-swizzled by the row
  const int tr_row = 4 * hi + (i16 >> 2);
  const int tr_clo = 2 * g16 + ((i16 & 3) >> 1);
  const int tr_byte = ((i16 & 3) & 1) * 8;
  const int tr_s1 = u_swz<D>(tr_row), tr_s2 = u_swz<D>(tr_row + 8);
  const int tr_b1 = tr_row * (D * 2) + tr_byte, tr_b2 = (tr_row + 8) * (D * 2) + tr_byte;
  const float sc = p.scale_log2;

  if (nu > 0) dma_issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == BWD_DQ && fuse) {
    float d = 0.f;
#pragma unroll
    for (int s = 0; s < DS; ++s) d = Dot8<T>::f(r2f[s], of[s], d);
    d += __shfl_xor(d, 32, 64);                       // the row's other half of the columns
    delta_lane = d;
    if (hi == 0 && my_row < p.Nq) p.delta_w[stat_i] = d;
  }
  // dQ launch, two waves per SIMD: dP starts at -delta of the lane's row instead of 0 — sixteen registers that never change, read as the C operand of each
  // half's first dP MFMA where it read the inline constant 0 — and dS = P o dP' needs no subtraction per element (32 VALU per tile; every path of the
  // launch — the hand-scheduled tiles, the compiler-scheduled ones, the windowed instantiation — forms the same sums in the same order)
  constexpr bool DQ_CINIT = MODE == BWD_DQ && NW == 8 && D == 128;   // (64 wide: the sixteen registers would cost the kernel its second workgroup per CU — 142 instead of 126)
  f32x16 dinit;
  if constexpr (DQ_CINIT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dinit[r] = -delta_lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(dinit[r]));   // (a register array, not re-materialised per use)
  }
  // (one wave per SIMD, head dims above 128: the resident fragments live in AccVGPRs — the MFMAs read them there; pinned in
  //  architectural VGPRs hipcc parks them in AccVGPRs anyway and copies four dwords back in front of every MFMA)
#pragma unroll
  for (int s = 0; s < DS; ++s) {
    if (NW == 4) asm volatile("" : "+a"(r1f[s])); else asm volatile("" : "+v"(r1f[s]));
  }
  if (NEED_DP) {
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      if (NW == 4) asm volatile("" : "+a"(r2f[s])); else asm volatile("" : "+v"(r2f[s]));
    }
  }
  asm volatile("s_barrier" ::: "memory");

  // ---- the wave's unmasked, whole tiles 0 .. a_uend - 1, hand-scheduled (tile u + 1 must exist and lie wholly inside the keys: its LDS-DMA pieces take the tile's
  //      byte offset as the SCALAR offset, which the descriptor's bounds check ignores); the loop below takes over at tile u0 with the same stage / barrier protocol
  int u0 = 0;
  if constexpr (ASMDQ) {
    // tiles 0 .. a_uend - 1 need no mask, a_uend .. a_mend - 1 are the wave's diagonal tiles (its masked bodies); every tile worked on or requested inside the
    // statement lies wholly inside the keys: all of the block's when the last one is whole, else up to the last but one
    int lim_t = nu;
    if (p.Nk % BN != 0) { lim_t = nu - 1; const int whole = p.Nk / BN - 1; lim_t = whole < lim_t ? whole : lim_t; }
    int a_uend = lim_t, a_mend = lim_t;
    if (CAUSAL) {
      const int d = wave_row0 + shift - (BN - 1);            // last key of tile u <= the limit of the wave's first row: u <= d / 64
      const int nun = d >= 0 ? d / BN + 1 : 0;
      const int e = wave_row0 + 31 + shift;                  // first key of tile u <= the limit of the wave's last row: u <= e / 64
      const int nac = e >= 0 ? e / BN + 1 : 0;
      a_uend = nun < a_uend ? nun : a_uend;
      a_mend = nac < a_mend ? nac : a_mend;
    }
    a_uend = __builtin_amdgcn_readfirstlane(a_uend);
    a_mend = __builtin_amdgcn_readfirstlane(a_mend);
    if (a_mend > 0) {
      f32x16 as0, ap0, as1, ap1;
      u32x4 af0, af1, af2, af3, aka;
      unsigned aka5, aka6, aka7;
      const unsigned a_kaddr = lds_base + (unsigned)k_rd_base + ((unsigned)(hi ^ k_rd_swz) << 4);
      const unsigned a_vat = lds_base + 4u * TILE_BYTES + (unsigned)v_rd_base;
      const unsigned a_ldsw = __builtin_amdgcn_readfirstlane(lds_base + wave * (PPW * 1024));
      const int a_kstr = __builtin_amdgcn_readfirstlane(tile_stride[0]), a_vstr = __builtin_amdgcn_readfirstlane(tile_stride[1]);
      const float a_sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc)));
      const int a_nu = __builtin_amdgcn_readfirstlane(nu);
      int a_lim = my_row + shift - a_uend * BN - 4 * hi;     // the masked bodies: keys of the tile beyond this offset lie behind the lane's row
      int a_ts;
      unsigned long long a_msk;
      float a_ninf;
      int ua = 0, koff = a_kstr, voff = a_vstr;
      if constexpr (std::is_same<T, __bf16>::value) { TFA_BWD_DQ_ASM_STMT(TFA_BWD_DQ_ASM_LOOP); }
      else { TFA_BWD_DQ_ASM_STMT(TFA_BWD_DQ_ASM_LOOP_F16); }
      u0 = ua;
    }
  }

#pragma nounroll
  for (int u = u0; u < nu; ++u) {
    const int stage = u & 1;
    if (u + 1 < nu) dma_issue(u + 1, stage ^ 1);     // the other stage was released by the barrier that ended tile u-1
    const int g = KEYS_RES ? u / ntl : 0;
    const int jt = t_begin + (u - g * ntl);
    const int row0 = jt * BN;                        // first streamed row of the tile (a key for dQ, a query for dK/dV)
    const char* img0 = smem + img_off(stage, 0);
    const char* img1 = smem + img_off(stage, 1);
    const char* imgt = smem + img_off(stage, IMG_TR);

    // does any element of this wave's 32 x 64 piece need masking?  is all of it masked (then the wave only keeps
    // the DMA and the barrier going)?
    bool need_mask, active = true;
    if (!KEYS_RES) {
      need_mask = (row0 + BN > p.Nk);
      if (CAUSAL) {
        need_mask = need_mask || (row0 + BN - 1 > wave_row0 + shift);
        active = row0 <= wave_row0 + 31 + shift;
      }
    } else {
      need_mask = CAUSAL && (row0 < wave_row0 + 31 - shift);
      if (CAUSAL) active = row0 + BN - 1 >= wave_row0 - shift;
    }
    const char* const st_img = smem + 2 * NIMG * TILE_BYTES + stage * 512;   // this tile's LSE (256 B) and delta (256 B)

    X8 pk[4];
    if (active) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // ---- GEMM-I: S (and dP) for the 32 tile rows of half t ----------------------------------------------------
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = DQ_CINIT ? dinit[r] : 0.f; }
      if constexpr (NW == 4) {
        // resident fragments in AccVGPRs, S / dP in VGPRs (Elem::mfma_bacc).  The MFMAs are asm statements, which hipcc keeps in
        // program order and does not pipeline LDS reads around: the fragments are read in groups of eight k-steps, one group ahead.
        constexpr int GK = DVB, NG = 2;                        // two groups of DVB k-steps
        X8 fa[2][GK], fb[NEED_DP ? 2 : 1][NEED_DP ? GK : 1];
        auto rd = [&](int g, int buf) {
#pragma unroll
          for (int i = 0; i < GK; ++i) {
            const int off = k_rd_base + t * 32 * (D * 2) + (((2 * (g * GK + i) + hi) ^ k_rd_swz) << 4);
            fa[buf][i] = __builtin_bit_cast(X8, lds_read_b128(img0, off));
            if (NEED_DP) fb[buf][i] = __builtin_bit_cast(X8, lds_read_b128(img1, off));
          }
        };
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) rd(g + 1, (g + 1) & 1);
#pragma unroll
          for (int i = 0; i < GK; ++i) {
            if (g == 0 && i == 0) {                  // (s and dp were just zeroed by VALU moves: the padded form)
              E::template mfma_bacc<true>(fa[0][0], r1f[0], s);
              if (NEED_DP) E::template mfma_bacc<true>(fb[0][0], r2f[0], dp);
            } else {
              E::template mfma_bacc<false>(fa[g & 1][i], r1f[g * GK + i], s);
              if (NEED_DP) E::template mfma_bacc<false>(fb[g & 1][i], r2f[g * GK + i], dp);
            }
          }
        }
        mfma_drain(s);
        if (NEED_DP) asm volatile("" : "+v"(dp));    // (dP's last MFMA sits in front of the drain as well: asm statements keep their order)
      } else {
#pragma unroll
        for (int sl = 0; sl < DS; ++sl) {
          const int off = k_rd_base + t * 32 * (D * 2) + (((2 * sl + hi) ^ k_rd_swz) << 4);
          s = E::mfma(__builtin_bit_cast(X8, lds_read_b128(img0, off)), r1f[sl], s);
          if (NEED_DP) dp = E::mfma(__builtin_bit_cast(X8, lds_read_b128(img1, off)), r2f[sl], dp);
        }
      }
      // ---- statistics of the 16 tile rows this lane sees (dK/dV) --------------------------------------------------
      float lse2[16], dl[16];
      if (KEYS_RES) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int q = row0 + 32 * t + 8 * g4 + 4 * hi;
          const f32x4 a = *reinterpret_cast<const f32x4*>(st_img + (q - row0) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) lse2[4 * g4 + e] = a[e] * 1.4426950408889634f;
          if (NEED_DP) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(st_img + 256 + (q - row0) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dl[4 * g4 + e] = c[e];
          }
        }
      }
      // ---- mask ---------------------------------------------------------------------------------------------------
      if (need_mask) {
        if (!KEYS_RES) {
          int lim = p.Nk - 1;
          if (CAUSAL) { const int c = my_row + shift; lim = c < lim ? c : lim; }
          lim -= row0 + 4 * hi;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ko = 32 * t + (r & 3) + 8 * (r >> 2);
            if (ko > lim) s[r] = -INFINITY;
          }
        } else {
          const int limq = my_row - shift - row0 - 4 * hi;     // query offsets below this do not see the lane's key
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qo = 32 * t + (r & 3) + 8 * (r >> 2);
            if (qo < limq) s[r] = -INFINITY;
          }
        }
      }
      // ---- P, dS, 16-bit pack ---------------------------------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float l2 = KEYS_RES ? lse2[r] : lse2_lane;
        const float pr = fast_exp2(fmaf(s[r], sc, -l2));
        float y = pr;
        if (NEED_DP) y = DQ_CINIT ? pr * dp[r] : pr * (dp[r] - (KEYS_RES ? dl[r] : delta_lane));
        pk[t * 2 + (r >> 3)][r & 7] = (T)y;
      }
      // ---- GEMM-II for the two 16-row slots of this half -----------------------------------------------------------
      auto rd_tr = [&](int sl, int d) -> X8 {
        s16x4 lo, hh;
        if (UNI) {
          const int c = 4 * d + tr_clo;
          lo = lds_read_tr16_b64(imgt + sl * 16 * (D * 2) + tr_b1 + ((c ^ tr_s1) << 4));
          hh = lds_read_tr16_b64(imgt + sl * 16 * (D * 2) + tr_b2 + ((c ^ tr_s2) << 4));
        } else {
          const char* a = imgt + v_rd_base + (sl * 2 * DT_L << 9) + (d << 9);
          lo = lds_read_tr16_b64(a);
          hh = lds_read_tr16_b64(a + 256);
        }
        s16x8 vf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(X8, vf);
      };
      if constexpr (OWN_ACC) {
        // (asm MFMAs again: the DT transposed fragments of a 16-row slot are read one slot ahead of the MFMAs that use them)
        X8 vt[2][DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) vt[0][d] = rd_tr(2 * t, d);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k == 0) {
#pragma unroll
            for (int d = 0; d < DT; ++d) vt[1][d] = rd_tr(2 * t + 1, d);
          }
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            if (d == 0) g_mfma_d<T, true>(d, vt[k][d], pk[2 * t + k]);
            else g_mfma_d<T, false>(d, vt[k][d], pk[2 * t + k]);
          }
        }
      } else {
#pragma unroll
        for (int sl = 2 * t; sl < 2 * t + 2; ++sl)
#pragma unroll
          for (int d = 0; d < DT; ++d) acc[d] = E::mfma(rd_tr(sl, d), pk[sl], acc[d]);
      }
    }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- epilogue: acc[dt][r] = grad[row my_row][32*dt + (r&3) + 8*(r>>2) + 4*hi] ----------------------------------------
  const float osc = (MODE == BWD_DV) ? 1.f : p.scale;
  float gout[OWN_ACC ? DT : 1][16];                  // OWN_ACC: the accumulators read out of a[0:127]
  if constexpr (OWN_ACC) {
#pragma unroll
    for (int d = 0; d < DT; ++d) g_read_d(d, gout[d]);
  }
  auto ga = [&](int d, int r) -> float { return OWN_ACC ? gout[OWN_ACC ? d : 0][r] : acc[OWN_ACC ? 0 : d][r]; };
  if (F32OUT) {
    float* gb = reinterpret_cast<float*>(p.grad) + b * p.gs_b + hr * p.gs_h;
    auto g_rs = BIG ? rsrc_at(gb, p.g_full, (unsigned long long)r0 * (unsigned long long)p.gs_n * 4ull) : __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, p.g_bytes, 0x00020000);
    const int goff = (my_row - (BIG ? r0 : 0)) * (int)p.gs_n * 4 + hi * 16;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v4 = {ga(d, 4 * g4 + 0) * osc, ga(d, 4 * g4 + 1) * osc, ga(d, 4 * g4 + 2) * osc, ga(d, 4 * g4 + 3) * osc};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), g_rs, d * 32 + g4 * 8 + hi * 4 < p.dv ? goff + (d * 32 + g4 * 8) * 4 : (int)TFA_OOB, 0, 0);
      }
  } else {
    // Whole rows through LDS instead of 16 eight-byte stores per lane over 32 different rows (tfa_bwd_kv_kernel.h's epilogue, the forward's
    // tfa_fwd_il_epilogue.inc): the wave transposes its 32 x D tile through its own slice of the idle tile stages (16-byte chunk index XOR
    // row) and stores 16 bytes per lane, 1 KiB contiguous per instruction.  Every wave is behind the tile loop's last barrier.
    T* gb = reinterpret_cast<T*>(p.grad) + b * p.gs_b + hr * p.gs_h;
    auto g_rs = BIG ? rsrc_at(gb, p.g_full, (unsigned long long)r0 * (unsigned long long)p.gs_n * 2ull) : __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, p.g_bytes, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) T t4;
    static_assert(NW * 32 * D * 2 <= 2 * NIMG * TILE_BYTES, "one 32 x D slice per wave inside the tile stages");
    // (the lane ids go through an empty asm: nothing below is computed in front of the tile loop and kept live across it)
    int qix = qi, lanex = lane, hix = hi;
    asm volatile("" : "+v"(qix), "+v"(lanex), "+v"(hix));
    char* const ow = smem + wave * (32 * D * 2);
    constexpr int CH = D / 8;                        // 16-byte chunks per row
    const int osw = (CH >= 16) ? (qix & 15) : (qix & 7);
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        t4 v4 = {(T)(ga(d, 4 * g4 + 0) * osc), (T)(ga(d, 4 * g4 + 1) * osc), (T)(ga(d, 4 * g4 + 2) * osc), (T)(ga(d, 4 * g4 + 3) * osc)};
        const int c = d * 4 + g4;
        *reinterpret_cast<u32x2*>(ow + qix * (D * 2) + ((c ^ osw) << 4) + hix * 8) = __builtin_bit_cast(u32x2, v4);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no barrier needed
    constexpr int RPI = 64 / CH;                     // rows per store instruction (2 at D = 256, 4 at D = 128, 8 at D = 64)
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int r = i * RPI + lanex / CH, cpos = lanex % CH;
      const int c = cpos ^ ((CH >= 16) ? (r & 15) : (r & 7));
      const u32x4 v = *reinterpret_cast<const u32x4*>(ow + r * (D * 2) + (cpos << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, g_rs, c * 8 < p.dv ? (wave_row0 - (BIG ? r0 : 0) + r) * (int)p.gs_n * 2 + (c << 4) : (int)TFA_OOB, 0, 0);
    }
  }
}

// delta[b,h,i] = sum_d dO[b,h,i,d] * O[b,h,i,d]  (fp32).  LPR lanes per row, 8 elements (16 bytes) per lane.
template <typename T, int D>
__global__ __launch_bounds__(256) void bwd_delta_kernel(const void* o, const void* dout, float* delta, long long os_b, long long os_h, long long os_n,
                                                       long long ds_b, long long ds_h, long long ds_n, int H, int Nq, long long rows, int dv) {
  constexpr int LPR = D / 8;
  typedef __attribute__((ext_vector_type(8))) T t8;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = gid / LPR;
  const int c = (int)(gid % LPR);
  float acc = 0.f;
  if (row < rows && c * 8 < dv) {
    const long long bh = row / Nq;
    const int i = (int)(row - bh * Nq);
    const int b = (int)(bh / H), h = (int)(bh - (long long)b * H);
    const t8 a = *reinterpret_cast<const t8*>(reinterpret_cast<const T*>(o) + b * os_b + h * os_h + i * os_n + c * 8);
    const t8 g = *reinterpret_cast<const t8*>(reinterpret_cast<const T*>(dout) + b * ds_b + h * ds_h + i * ds_n + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)g[e];
  }
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (row < rows && c == 0) delta[row] = acc;
}

}  // namespace tfa
