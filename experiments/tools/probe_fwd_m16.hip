// experiments/tools/probe_fwd_m16.hip — PROTOTYPE: the forward tile math on v_mfma_f32_16x16x32_bf16 instead of 32x32x16.
//
// Why: under the board's power cap the 16x16x32 instruction sustains 13-16 % more flops than 32x32x16 on random operands
// (profiles/r03_mfma_power_16x16x32.txt).  This standalone program holds the smallest complete kernel that uses it — one shape
// family only (bf16, D = 128, contiguous (B*H, N, 128) tensors, N a multiple of 256, causal or not), compiler-scheduled builtin
// MFMAs, exact running max — to measure what the instruction buys in a whole kernel before the product kernel is re-tiled.
// It is NOT part of the library and nothing in the package uses it.
//
// Tiling (one wave = 32 query rows = two 16-row blocks rb; a KV tile = 64 keys = four 16-key blocks kb):
//   S^T[kb][rb] (16 keys x 16 rows, f32x4)  = sum over 4 k-steps of  A = K frag [16 keys x 32 d] (LDS, one ds_read_b128, feeds both rb)
//                                                                     B = Q frag [32 d x 16 rows] (registers, loaded once)
//   C layout: lane l holds rows(keys) 4*(l>>4)+r, column (query row) l&15.
//   P -> B operand of the PV MFMAs WITHOUT moving data: k-step j of the PV GEMM takes its 32 keys in the order
//     k = 8g+i  <->  key 32j + 4g + i (i < 4),  key 32j + 16 + 4g + (i-4) (i >= 4)        (g = l>>4)
//   i.e. a lane's B fragment is {P[kb=2j][rb][0..3], P[kb=2j+1][rb][0..3]}; V's transposed reads (ds_read_b64_tr_b16) deliver
//   the A operand V^T[16 d x 32 keys] in the same key order.
//   O^T[db][rb] (16 d x 16 rows, f32x4), db = 0..7.
// Per tile and wave: 32 + 32 MFMAs (16 KFLOP each), 16 ds_read_b128 + 32 ds_read_b64_tr_b16 — the LDS traffic of the 32x32x16 tiling.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tiny-flash-attention_amd/csrc -I include -o probe_fwd_m16 experiments/tools/probe_fwd_m16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
#include "tfa_fwd_kernel_dma.h"

using namespace tfa;
typedef __attribute__((ext_vector_type(4))) float f32x4_;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;

constexpr int HD = 128, BN = 64, NWV = 8, BM = 256, TILE_B = BN * HD * 2;

struct MArgs {
  const __bf16 *q, *k, *v;
  __bf16* o;
  float* lse;
  int N;
  float scale_log2;
  unsigned long long* tr;   // M16_TRACE builds: per wave {life, memory wait at the top of an iteration, barrier wait} in s_memtime ticks
};

static __device__ __forceinline__ float xor16(float x) {   // lane l <-> lane l^16 (ds_swizzle bit mode: and 31, or 0, xor 16)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
}
static __device__ __forceinline__ float red_max4(float x) {   // over the four 16-lane groups that share a column
  x = fmaxf(x, xor16(x));
  return pair_max(x);
}
static __device__ __forceinline__ float red_sum4(float x) {
  x = x + xor16(x);
  return pair_sum(x);
}

template <bool CAUSAL>
__global__ __launch_bounds__(NWV * 64, 2) void fwd_m16(const MArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // K[2], V[2] tile buffers
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int nqb = p.N / BM;
  const int bh = blockIdx.y;
  const int qb = CAUSAL ? nqb - 1 - (int)blockIdx.x : (int)blockIdx.x;   // causal: the block with the most tiles first
  const int q0 = qb * BM + wave * 32;
  const size_t head = (size_t)bh * p.N * HD;
  const int nt = CAUSAL ? (qb * BM + BM) / BN : p.N / BN;
  const int nt_wave = CAUSAL ? (q0 + 32 + BN - 1) / BN : nt;             // tiles this wave computes

  auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + head), 0, (unsigned)(p.N * HD * 2), 0x00020000);
  auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v + head), 0, (unsigned)(p.N * HD * 2), 0x00020000);
  // LDS-DMA pieces: piece pc = rows 4pc .. 4pc+3 of the tile, lane -> (row 4pc + lane/16, chunk position lane%16); the SOURCE chunk
  // is the position XOR the row's swizzle (K: row & 15 at 16-byte granularity; V: (row & 7) << 1, i.e. 32-byte granularity)
  int srcK[2], srcV[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave * 2 + i, row = pc * 4 + (lane >> 4), cpos = lane & 15;
    srcK[i] = row * (HD * 2) + ((cpos ^ (row & 15)) << 4);
    srcV[i] = row * (HD * 2) + ((cpos ^ ((row & 7) << 1)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) lds_dma16_m0(rsK, lds_base + (wave * 2 + i) * 1024, srcK[i]);   // K(0); V(0) follows at the top of iteration 0

  // Q fragments: B operand [32 d x 16 rows]: lane (n, g) holds d = 32 ks + 8 g + 0..7 of row q0 + 16 rb + n
  bf16x8 qf[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[rb][ks] = *reinterpret_cast<const bf16x8*>(p.q + head + (size_t)(q0 + rb * 16 + n) * HD + 32 * ks + 8 * g);

  f32x4_ o[8][2];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) o[db][rb] = f32x4_{0.f, 0.f, 0.f, 0.f};
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};   // m in log2 units; l = this lane's PARTIAL row sum (reduced at the end)
  const float sc = p.scale_log2;

  // per-lane LDS read offsets
  const int k_rd = n * (HD * 2);                               // + kb*16 rows; chunk (4ks + g) ^ (key & 15), key & 15 == n
  const int v_key = 4 * g + (n >> 2);                          // + 32 j + 16 h
  const int v_c16 = (n & 3) >> 1, v_byte = ((n & 3) & 1) * 8;  // 16-byte chunk 2 db + v_c16, XOR ((key & 7) << 1)

  // One tile behind: iteration t computes S(t) and its softmax while the matrix pipe works on O += V(t-1) P(t-1) — two independent
  // instruction streams in one basic block, interleaved by the compiler's scheduler.  K(t+1) and V(t) are requested at the top of
  // iteration t (K(t-1)'s and V(t-2)'s buffers are free: every wave passed the barrier).
  bf16x8 pb[2][2];
  auto pv = [&](int tv) {
    const char* vb_ = smem + (2 + (tv & 1)) * TILE_B;
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int key0 = 32 * j + v_key, key1 = key0 + 16;
        const s16x4 lo = lds_read_tr16_b64(vb_ + key0 * (HD * 2) + (((2 * db + v_c16) ^ ((key0 & 7) << 1)) << 4) + v_byte);
        const s16x4 hh = lds_read_tr16_b64(vb_ + key1 * (HD * 2) + (((2 * db + v_c16) ^ ((key1 & 7) << 1)) << 4) + v_byte);
        const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7));
        o[db][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[j][0], o[db][0], 0, 0, 0);
        o[db][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[j][1], o[db][1], 0, 0, 0);
      }
  };
  auto issue_kv = [&](int t) {                                 // K(t+1) -> K buffer (t+1)&1, V(t) -> V buffer t&1
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (t + 1 < nt) lds_dma16_m0(rsK, lds_base + ((t + 1) & 1) * TILE_B + (wave * 2 + i) * 1024, srcK[i] + (t + 1) * TILE_B);
      lds_dma16_m0(rsV, lds_base + (2 + (t & 1)) * TILE_B + (wave * 2 + i) * 1024, srcV[i] + t * TILE_B);
    }
  };
  auto tile = [&](int t, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;           // t == 0: nothing to add to O yet
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    issue_kv(t);
    const bool act = t < nt_wave;                              // (causal: this wave's rows end before this tile)
    if (!act) {
      if (t == nt_wave) pv(t - 1);
      return;
    }
    const char* kb_ = smem + (t & 1) * TILE_B;

    // ---- S^T = K Q^T -------------------------------------------------------------------------------------------
    f32x4_ s[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      s[kb][0] = f32x4_{0.f, 0.f, 0.f, 0.f};
      s[kb][1] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, lds_read_b128(kb_, kb * 16 * (HD * 2) + k_rd + (((4 * ks + g) ^ n) << 4)));
        s[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[0][ks], s[kb][0], 0, 0, 0);
        s[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[1][ks], s[kb][1], 0, 0, 0);
      }
    }
    if (CAUSAL && t * BN + BN - 1 > q0) {                      // the wave's diagonal tiles
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * BN + kb * 16 + 4 * g + r > q0 + rb * 16 + n) s[kb][rb][r] = -INFINITY;
    }
    // ---- O += V(t-1) P(t-1) on the matrix pipe, softmax of tile t on the VALU ---------------------------------------
    if constexpr (!FIRST) pv(t - 1);
    bf16x8 pn[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float mx = s[0][rb][0];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][rb][r]);
      mx = red_max4(mx) * sc;
      const float m_new = fmaxf(m[rb], mx);
      float pr[4][4], ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[kb][r] = fast_exp2(fmaf(s[kb][rb][r], sc, -m_new));
          ps += pr[kb][r];
        }
      {                                                        // rescale O (tiles 0..t-1) and l: unconditional — a branch here would split
        const float alpha = fast_exp2(m[rb] - m_new);          // the basic block and the scheduler could not interleave across it
        l[rb] *= alpha;
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[db][rb][r] *= alpha;
        m[rb] = m_new;
      }
      l[rb] += ps;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b[r] = (__bf16)pr[2 * j][r];
          b[4 + r] = (__bf16)pr[2 * j + 1][r];
        }
        pn[j][rb] = b;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { pb[j][0] = pn[j][0]; pb[j][1] = pn[j][1]; }
#if defined(M16_SCHED)
    // scheduling hint for the block above: after the 32 S MFMAs, 32 groups of {1 MFMA, 2 LDS reads, M16_SCHED VALU}
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // 2 DS reads
      __builtin_amdgcn_sched_group_barrier(0x002, M16_SCHED, 0);   // VALU
    }
#endif
  };
  tile(0, std::true_type{});
#pragma nounroll
  for (int t = 1; t < nt; ++t) tile(t, std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // V(nt-1) has landed
  if (nt_wave == nt) pv(nt - 1);

  // ---- epilogue: O = O^T / l, LSE = m ln2 + ln l ----------------------------------------------------------------------
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float lt = red_sum4(l[rb]);
    const float inv = 1.f / lt;
    const int row = q0 + rb * 16 + n;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      bf16x4_ w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (__bf16)(o[db][rb][r] * inv);
      *reinterpret_cast<bf16x4_*>(p.o + head + (size_t)row * HD + db * 16 + 4 * g) = w;
    }
    if (g == 0) p.lse[(size_t)bh * p.N + row] = m[rb] * 0.6931471805599453f + logf(lt);
  }
}


// ---- the three-stage form (non-causal): iteration t issues the MFMAs of S(t) and of O += V(t-2) P(t-2) and, between them, the VALU
// work of softmax(S(t-1)) — three independent streams, interleaved BY HAND in 16 chunks (a sched_barrier after each keeps hipcc from
// re-clustering them).  Lazy row reference: P = exp2(S*scale - ref) with ref moved only when a row's max exceeds it by more than
// TH = 8 (P <= 256); the rescale of O and l is a rarely taken branch behind the block.
#if !defined(M16_TH)
#define M16_TH 8.0f
#endif
__global__ __launch_bounds__(NWV * 64, 2) void fwd_m16p(const MArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // K[2], V[2] tile buffers
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, qb = blockIdx.x;
  const int q0 = qb * BM + wave * 32;
  const size_t head = (size_t)bh * p.N * HD;
  const int nt = p.N / BN;
  auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + head), 0, (unsigned)(p.N * HD * 2), 0x00020000);
  auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v + head), 0, (unsigned)(p.N * HD * 2), 0x00020000);
  int srcK[2], srcV[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave * 2 + i, row = pc * 4 + (lane >> 4), cpos = lane & 15;
    srcK[i] = row * (HD * 2) + ((cpos ^ (row & 15)) << 4);
    srcV[i] = row * (HD * 2) + ((cpos ^ ((row & 7) << 1)) << 4);
  }
  auto dmaK = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_dma16_m0(rsK, lds_base + (t & 1) * TILE_B + (wave * 2 + i) * 1024, srcK[i] + t * TILE_B);
  };
  auto dmaV = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_dma16_m0(rsV, lds_base + (2 + (t & 1)) * TILE_B + (wave * 2 + i) * 1024, srcV[i] + t * TILE_B);
  };
  dmaK(0);
  bf16x8 qf[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[rb][ks] = *reinterpret_cast<const bf16x8*>(p.q + head + (size_t)(q0 + rb * 16 + n) * HD + 32 * ks + 8 * g);
  f32x4_ o[8][2];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) o[db][rb] = f32x4_{0.f, 0.f, 0.f, 0.f};
  float ref[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  const float sc = p.scale_log2;
  const int k_rd = n * (HD * 2);
  const int v_key = 4 * g + (n >> 2);
  const int v_c16 = (n & 3) >> 1, v_byte = ((n & 3) & 1) * 8;

  auto k_frag = [&](const char* kb_, int kb, int ks) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, lds_read_b128(kb_, kb * 16 * (HD * 2) + k_rd + (((4 * ks + g) ^ n) << 4)));
  };
  auto v_frag = [&](const char* vb_, int db, int j) -> bf16x8 {
    const int key0 = 32 * j + v_key, key1 = key0 + 16;
    const s16x4 lo = lds_read_tr16_b64(vb_ + key0 * (HD * 2) + (((2 * db + v_c16) ^ ((key0 & 7) << 1)) << 4) + v_byte);
    const s16x4 hh = lds_read_tr16_b64(vb_ + key1 * (HD * 2) + (((2 * db + v_c16) ^ ((key1 & 7) << 1)) << 4) + v_byte);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto qk_all = [&](const char* kb_, f32x4_ (&s)[4][2]) {       // plain S = K Q^T (prologue)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      s[kb][0] = f32x4_{0.f, 0.f, 0.f, 0.f};
      s[kb][1] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = k_frag(kb_, kb, ks);
        s[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[0][ks], s[kb][0], 0, 0, 0);
        s[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[1][ks], s[kb][1], 0, 0, 0);
      }
    }
  };
  // row maxima of a finished S tile (log2 units), reduced over the four lane groups of a column
  auto row_max = [&](const f32x4_ (&s)[4][2], float (&mx)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float x = s[0][rb][0];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) x = fmaxf(x, s[kb][rb][r]);
      mx[rb] = red_max4(x) * sc;
    }
  };

  // ---- prologue: S(0), S(1) ---------------------------------------------------------------------------------------------
  f32x4_ sP[4][2], sC[4][2];                                   // sP = S(t-1) (finished), sC = S(t) (accumulating)
  bf16x8 pb[2][2];                                             // P(t-2) as PV B fragments
#pragma unroll
  for (int j = 0; j < 2; ++j) { pb[j][0] = bf16x8{}; pb[j][1] = bf16x8{}; }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  dmaK(1);
  qk_all(smem, sP);                                            // S(0)
  // iterations t = 1 .. nt+1: QK(t) if t < nt; softmax(S(t-1)) if t-1 < nt; PV(t-2) if t >= 2
#if defined(M16_TRACE)
  unsigned long long tw_mem = 0, tw_bar = 0;
  const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
  auto top = [&](int t) {
#if defined(M16_TRACE)
    const unsigned long long a0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long a1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_barrier" ::: "memory");
    tw_mem += a1 - a0; tw_bar += __builtin_amdgcn_s_memtime() - a1;
#else
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // K(t), V(t-2) landed; everyone is done with iteration t-1's buffers
#endif
    if (t + 1 < nt) dmaK(t + 1);                               // into K(t-1)'s buffer
    if (t - 1 < nt) dmaV(t - 1);                               // into V(t-3)'s buffer; read in iteration t+1
  };
  auto hot = [&](int t, f32x4_ (&sPv)[4][2], f32x4_ (&sCu)[4][2]) {   // sPv = S(t-1), finished; sCu receives S(t)
    top(t);
    const char* kb_ = smem + (t & 1) * TILE_B;
    const char* vb_ = smem + (2 + ((t - 2) & 1)) * TILE_B;
    // ================= the steady state: one basic block, 16 chunks of 4 MFMAs =================
    // chunks 0-3 also carry the row maxima of S(t-1) (in-lane max, two cross-lane steps, the new references), chunks 4-11 its 32
    // exponentials (four per chunk, packed to bf16 as they appear), chunks 12-15 nothing but MFMAs
    float mxl[2], nref[2], alpha[2], ps[2] = {0.f, 0.f};
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    unsigned pn32[2][2][4];                                    // P(t-1): [rb][j][dword of the 8-element B fragment]
#if !defined(M16_PF)
#define M16_PF 1                                               // chunks of LDS prefetch distance
#endif
    bf16x8 kq[M16_PF + 1], vq[M16_PF + 1];                     // fragment queues (static indices after unrolling)
#pragma unroll
    for (int i = 0; i < M16_PF; ++i) { kq[i] = k_frag(kb_, i >> 2, i & 3); vq[i] = v_frag(vb_, i >> 1, i & 1); }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int kb = c >> 2, ks = c & 3;                       // QK chunk: K fragment (kb, ks) -> both row blocks
      const int db = c >> 1, j = c & 1;                        // PV chunk: V fragment (db, j) -> both row blocks
      const bf16x8 ka_use = kq[c % (M16_PF + 1)], va_use = vq[c % (M16_PF + 1)];
      if (c + M16_PF < 16) {
        kq[(c + M16_PF) % (M16_PF + 1)] = k_frag(kb_, (c + M16_PF) >> 2, (c + M16_PF) & 3);
        vq[(c + M16_PF) % (M16_PF + 1)] = v_frag(vb_, (c + M16_PF) >> 1, (c + M16_PF) & 1);
      }
      const f32x4_ zero = {0.f, 0.f, 0.f, 0.f};                // (an inline-constant C operand: no register to clear)
      sCu[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka_use, qf[0][ks], ks == 0 ? zero : sCu[kb][0], 0, 0, 0);
      sCu[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka_use, qf[1][ks], ks == 0 ? zero : sCu[kb][1], 0, 0, 0);
      o[db][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_use, pb[j][0], o[db][0], 0, 0, 0);
      o[db][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_use, pb[j][1], o[db][1], 0, 0, 0);
      if (c == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          float x = sPv[0][rb][0];
#pragma unroll
          for (int kb2 = 0; kb2 < 4; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) x = fmaxf(x, sPv[kb2][rb][r]);
          mxl[rb] = x;
          asm volatile("" : "+v"(mxl[rb]));
        }
      } else if (c == 1) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) { mxl[rb] = fmaxf(mxl[rb], xor16(mxl[rb])); asm volatile("" : "+v"(mxl[rb])); }
      } else if (c == 2) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) { mxl[rb] = pair_max(mxl[rb]) * sc; asm volatile("" : "+v"(mxl[rb])); }
      } else if (c == 3) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          nref[rb] = mxl[rb] > ref[rb] + M16_TH ? mxl[rb] : ref[rb];
          alpha[rb] = fast_exp2(ref[rb] - nref[rb]);
          asm volatile("" : "+v"(nref[rb]), "+v"(alpha[rb]));
        }
      } else if (c < 12) {
        // four elements per lane: e = 4 (c - 4) .. +3  ->  (kb', rb', r) = (e >> 3, (e >> 2) & 1, e & 3)
        const int e0i = 4 * (c - 4), kb2 = e0i >> 3, rb2 = (e0i >> 2) & 1;
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = fast_exp2(fmaf(sPv[kb2][rb2][r], sc, -nref[rb2]));
        ps[rb2] += (e[0] + e[1]) + (e[2] + e[3]);
        const bf16x2_ w0 = {(__bf16)e[0], (__bf16)e[1]}, w1 = {(__bf16)e[2], (__bf16)e[3]};
        unsigned u0 = __builtin_bit_cast(unsigned, w0), u1 = __builtin_bit_cast(unsigned, w1);
        asm volatile("" : "+v"(u0), "+v"(u1), "+v"(ps[rb2]));  // pin: hipcc sinks work whose result is only used behind the block
        pn32[rb2][kb2 >> 1][(kb2 & 1) * 2 + 0] = u0;
        pn32[rb2][kb2 >> 1][(kb2 & 1) * 2 + 1] = u1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the PV MFMAs above added P(t-2) (old reference): rescale BEHIND them when a reference moved (rare)
    if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.f || alpha[1] != 1.f) != 0) {
      float a_[2] = {alpha[0], alpha[1]};
      asm volatile("" : "+v"(a_[0]), "+v"(a_[1]));             // (keeps hipcc from speculating the 64 multiplies into the hot block)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[db][rb][r] *= a_[rb];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      l[rb] = fmaf(l[rb], alpha[rb], ps[rb]);
      ref[rb] = nref[rb];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4 w = {pn32[rb][j][0], pn32[rb][j][1], pn32[rb][j][2], pn32[rb][j][3]};
        pb[j][rb] = __builtin_bit_cast(bf16x8, w);
      }
    }
  };
  // ================= the ends of the stream (t = 1, nt, nt+1): the same work, stage by stage =================
  auto ends = [&](int t) {
    top(t);
    const char* kb_ = smem + (t & 1) * TILE_B;
    const char* vb_ = smem + (2 + ((t - 2) & 1)) * TILE_B;
    const bool do_qk = t < nt, do_sm = t - 1 < nt, do_pv = t >= 2;
    if (do_pv) {
#pragma unroll
      for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 a = v_frag(vb_, db, j);
          o[db][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[j][0], o[db][0], 0, 0, 0);
          o[db][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[j][1], o[db][1], 0, 0, 0);
        }
    }
    if (do_qk) qk_all(kb_, sC);
    if (do_sm) {
      float mx[2];
      row_max(sP, mx);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float nref = mx[rb] > ref[rb] + M16_TH ? mx[rb] : ref[rb];
        const float alpha = fast_exp2(ref[rb] - nref);
        float ps = 0.f, pr[4][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pr[kb][r] = fast_exp2(fmaf(sP[kb][rb][r], sc, -nref));
            ps += pr[kb][r];
          }
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[db][rb][r] *= alpha;
        l[rb] = fmaf(l[rb], alpha, ps);
        ref[rb] = nref;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 b;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            b[r] = (__bf16)pr[2 * j][r];
            b[4 + r] = (__bf16)pr[2 * j + 1][r];
          }
          pb[j][rb] = b;
        }
      }
    }
    if (do_qk) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) { sP[kb][0] = sC[kb][0]; sP[kb][1] = sC[kb][1]; }
    }
  };
  ends(1);
  {
    int t = 2;
#pragma nounroll
    for (; t + 1 < nt; t += 2) {                               // two tiles per trip: S(t-1) / S(t) swap registers instead of being copied
      hot(t, sP, sC);
      hot(t + 1, sC, sP);
    }
    if (t < nt) {
      hot(t, sP, sC);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) { sP[kb][0] = sC[kb][0]; sP[kb][1] = sC[kb][1]; }
    }
  }
  ends(nt);
  ends(nt + 1);
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float lt = red_sum4(l[rb]);
    const float inv = 1.f / lt;
    const int row = q0 + rb * 16 + n;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      bf16x4_ w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (__bf16)(o[db][rb][r] * inv);
      *reinterpret_cast<bf16x4_*>(p.o + head + (size_t)row * HD + db * 16 + 4 * g) = w;
    }
    if (g == 0) p.lse[(size_t)bh * p.N + row] = ref[rb] * 0.6931471805599453f + logf(lt);
  }
#if defined(M16_TRACE)
  if (p.tr && lane == 0) {
    unsigned long long* w = p.tr + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWV + wave) * 4;
    w[0] = __builtin_amdgcn_s_memtime() - tw0; w[1] = tw_mem; w[2] = tw_bar; w[3] = (unsigned long long)nt;
  }
#endif
}

static unsigned short f2bf(float x) {
  unsigned u; memcpy(&u, &x, 4);
  return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static float bf2f(unsigned short h) {
  unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4);
  return x;
}
static float gauss() {
  const float u1 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
  struct Cfg { const char* name; int B, H, N; bool causal; } cfgs[] = {
      {"cfg3 (B4 H32 N4096 causal)", 4, 32, 4096, true}, {"cfg3nc (B4 H32 N4096)", 4, 32, 4096, false}, {"cfg4 (B1 H16 N16384)", 1, 16, 16384, false}};
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
  const bool piped = argc > 2 && argv[2][0] == 'p';           // "p": the three-stage hand-interleaved kernel for the non-causal shapes
  hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_m16p), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_B);
  hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_m16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_B);
  hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_m16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_B);
  for (const Cfg& c : cfgs) {
    const size_t ne = (size_t)c.B * c.H * c.N * HD;
    std::vector<unsigned short> hq(ne), hk(ne), hv(ne), ho(ne);
    srand(7);
    // (one head's worth of random numbers repeated over the heads: the host generator is slow and the check reads head 0 and the last)
    const size_t per = (size_t)c.N * HD;
    for (size_t i = 0; i < per; ++i) { hq[i] = f2bf(0.5f * gauss()); hk[i] = f2bf(0.5f * gauss()); hv[i] = f2bf(0.5f * gauss()); }
    for (size_t h = 1; h < (size_t)c.B * c.H; ++h) {
      const size_t rot = (h * 977) % per;                     // each head a rotation of head 0's values
      for (size_t i = 0; i < per; ++i) { hq[h * per + i] = hq[(i + rot) % per]; hk[h * per + i] = hk[(i + 2 * rot) % per]; hv[h * per + i] = hv[(i + 3 * rot) % per]; }
    }
    __bf16 *dq, *dk, *dv, *dout;
    float* dlse;
    hipMalloc(&dq, ne * 2); hipMalloc(&dk, ne * 2); hipMalloc(&dv, ne * 2); hipMalloc(&dout, ne * 2);
    hipMalloc(&dlse, (size_t)c.B * c.H * c.N * 4);
    hipMemcpy(dq, hq.data(), ne * 2, hipMemcpyHostToDevice);
    hipMemcpy(dk, hk.data(), ne * 2, hipMemcpyHostToDevice);
    hipMemcpy(dv, hv.data(), ne * 2, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, ne * 2);
    unsigned long long* dtr = nullptr;
    const size_t ntr = (size_t)(c.N / BM) * c.B * c.H * NWV * 4;
#if defined(M16_TRACE)
    hipMalloc(&dtr, ntr * 8);
    hipMemset(dtr, 0, ntr * 8);
#endif
    MArgs a{dq, dk, dv, dout, dlse, c.N, (1.0f / sqrtf((float)HD)) * 1.4426950408889634f, dtr};
    const dim3 grid(c.N / BM, c.B * c.H), block(NWV * 64);
    auto launch = [&]() {
      if (c.causal) hipLaunchKernelGGL(fwd_m16<true>, grid, block, 4 * TILE_B, 0, a);
      else if (piped) hipLaunchKernelGGL(fwd_m16p, grid, block, 4 * TILE_B, 0, a);
      else hipLaunchKernelGGL(fwd_m16<false>, grid, block, 4 * TILE_B, 0, a);
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", c.name, hipGetErrorString(hipGetLastError())); return 1; }
    hipMemcpy(ho.data(), dout, ne * 2, hipMemcpyDeviceToHost);
    std::vector<float> hl((size_t)c.B * c.H * c.N);
    hipMemcpy(hl.data(), dlse, hl.size() * 4, hipMemcpyDeviceToHost);
    // check sampled rows of the first and the last head against an fp64 host reference
    double worst = 0, worst_l = 0;
    const int heads[2] = {0, c.B * c.H - 1};
    const int rows[8] = {0, 1, 17, 255, 256, c.N / 2 + 37, c.N - 2, c.N - 1};
    for (int hh : heads)
      for (int r : rows) {
        const size_t hb = (size_t)hh * per;
        const int nk = c.causal ? r + 1 : c.N;
        std::vector<double> sj(nk);
        double mx = -1e300;
        for (int j = 0; j < nk; ++j) {
          double acc = 0;
          for (int d = 0; d < HD; ++d) acc += (double)bf2f(hq[hb + (size_t)r * HD + d]) * (double)bf2f(hk[hb + (size_t)j * HD + d]);
          sj[j] = acc / sqrt((double)HD);
          mx = sj[j] > mx ? sj[j] : mx;
        }
        double den = 0;
        for (int j = 0; j < nk; ++j) { sj[j] = exp(sj[j] - mx); den += sj[j]; }
        for (int d = 0; d < HD; ++d) {
          double acc = 0;
          for (int j = 0; j < nk; ++j) acc += sj[j] * (double)bf2f(hv[hb + (size_t)j * HD + d]);
          const double diff = fabs(acc / den - (double)bf2f(ho[hb + (size_t)r * HD + d]));
          worst = diff > worst ? diff : worst;
        }
        const double dl = fabs((mx + log(den)) - (double)hl[(size_t)hh * c.N + r]);
        worst_l = dl > worst_l ? dl : worst_l;
      }
    printf("ARM_BEGIN %s\n", c.name); fflush(stdout);       // (tools/power_trace.py --cmd: power and clock per configuration)
    // timing: pre-condition, then batches of 20 launches between events
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e9, sum = 0; int nb = 0;
    const double flops = 4.0 * c.B * c.H * (double)c.N * c.N * HD * (c.causal ? 0.5 : 1.0);
    float total_ms = 0;
    while (total_ms < seconds * 1000.0) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      total_ms += ms;
      if (total_ms > seconds * 500.0) { best = ms / 20 < best ? ms / 20 : best; sum += ms / 20; ++nb; }   // second half only
    }
    printf("%-28s m16 prototype%s: %7.4f ms = %7.1f TFLOP/s (mean of the warm half; best %7.1f) | max|out - fp64| %.2e  max|dLSE| %.2e %s\n", c.name, (piped && !c.causal) ? " (three-stage)" : "",
           sum / nb, flops / (sum / nb * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12, worst, worst_l, (worst <= 1e-2 && worst_l <= 1e-3) ? "ok" : "WRONG");
#if defined(M16_TRACE)
    if (piped && !c.causal) {
      std::vector<unsigned long long> ht(ntr);
      hipMemcpy(ht.data(), dtr, ntr * 8, hipMemcpyDeviceToHost);
      double life = 0, mem = 0, bar = 0, tiles = 0;
      for (size_t i = 0; i < ntr; i += 4) { life += ht[i]; mem += ht[i + 1]; bar += ht[i + 2]; tiles += ht[i + 3]; }
      printf("     trace: per tile and wave %.0f ticks, of which %.0f (%.1f %%) waiting for memory and %.0f (%.1f %%) at the barrier\n", life / tiles, mem / tiles,
             100 * mem / life, bar / tiles, 100 * bar / life);
    }
#endif
    printf("ARM_END %s\n", c.name);
    fflush(stdout);
    hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(dlse);
  }
  return 0;
}
