// tfa_bwd_dq_kernel.h — dQ from the dS workspace: dQ = scale * dS . K with ONE GEMM (gfx950).
//
// When the caller lends tfa_bwd a workspace (tfa_bwd_params::workspace), the fused dK/dV launch (tfa_bwd_kv_kernel.h, WS) has
// already produced dS = P o (dP - delta) for every visible (query, key) pair, rounded to 16 bit as every kernel's GEMM consumes
// it, stored TRANSPOSED — keys as rows, queries along them, in blocks of 128 keys x 64 queries — because that is how its lanes hold it
// (lane = key).  The default dQ
// launch (tfa_bwd_kernel.h) recomputes S and dP to get the same numbers: 3 GEMM units; this kernel reads them back: 1 unit, bound
// by streaming the workspace once (B*H*Nq*Nk*2 bytes, half of that when causal).
//
// The transposition costs nothing: a tile dS^T[64 keys][256 queries] (four contiguous 8 KiB sub-blocks of the workspace) is staged by LDS-DMA into the forward's V-tile image
// ([8 keys][32 columns] sub-tiles) with the QUERY index in the role of V's head dim, and ds_read_b64_tr_b16 hands every lane the
// fragment "my query, 8 keys" — the B operand that P is in the forward's O^T += V^T P^T.  The A operand is K^T, read the same
// way from K's V-layout image.  Acc^T[d, query] += K^T[d, keys] . dS^T[keys, query]: the result layout is the one every
// epilogue of this library stores.  256 queries per workgroup (8 waves x 32), 64 keys per tile, three stages, the tile after next
// in flight behind a counted vmcnt.  Causal: a wave skips key tiles its 32 queries cannot see — exactly the region the producer
// may have left unwritten.
#pragma once
#include "tfa_bwd_kernel.h"

namespace tfa {

template <typename T, int D, bool CAUSAL, bool F32OUT>
__global__ __launch_bounds__(512, 2) void bwd_dq_ws_kernel(const BArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int NW = 8;
  constexpr int BM = NW * 32;                      // queries per workgroup
  constexpr int BN = 64;                           // keys per tile
  constexpr int DT = D / 32;
  constexpr int QT = BM / 32;                      // 32-query column tiles of the dS^T image
  constexpr int S_BYTES = BN * BM * 2;             // dS^T tile image
  constexpr int K_BYTES = BN * D * 2;              // K tile image
  constexpr int STAGE_BYTES = S_BYTES + K_BYTES;
  constexpr int NSTAGE = 3;
  constexpr int PPW_S = S_BYTES / 1024 / NW;       // 4
  constexpr int PPW_K = K_BYTES / 1024 / NW;       // 2 (D = 128), 1 (D = 64)
  static_assert(PPW_S * NW * 1024 == S_BYTES && PPW_K * NW * 1024 == K_BYTES && PPW_K >= 1, "");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31;
  const int hi = lane >> 5;

  const int G = p.H / p.Hk;
  int rb, bh;
  {
    const int nbh = p.B * p.H, id = blockIdx.x;    // a (b, head) stays on one XCD (K tiles re-read from its L2); causal: the block with the most tiles first
    if ((nbh & 7) == 0) {
      const int x = id & 7, q8 = id >> 3;
      bh = x + 8 * (q8 / p.nrb);
      rb = q8 % p.nrb;
    } else {
      bh = id / p.nrb;
      rb = id % p.nrb;
    }
    if (CAUSAL) rb = p.nrb - 1 - rb;
  }
  const int b = bh / p.H;
  const int h = bh - b * p.H;
  const int hk = h / G;
  const int shift = p.Nk - p.Nq;
  const int q0 = rb * BM;
  const int wave_q0 = q0 + wave * 32;
  const int my_row = wave_q0 + qi;

  int kv_end = p.Nk;
  if (CAUSAL) {
    const int lim = q0 + BM + shift;               // one past the last key any row of the block sees
    kv_end = lim < kv_end ? lim : kv_end;
  }
  const int nt = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;

  // ---- descriptors and per-lane DMA sources ---------------------------------------------------------------------------
  const T* slab = reinterpret_cast<const T*>(p.ws) + (long long)bh * p.ws_nk * p.ws_nq;
  auto s_rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (unsigned)((long long)p.ws_nk * p.ws_nq * 2), 0x00020000);
  const T* kbase = reinterpret_cast<const T*>(p.k.p) + b * p.k.s_b + hk * p.k.s_h;
  auto k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, p.k.bytes, 0x00020000);
  int s_src[PPW_S], k_src[PPW_K];
#pragma unroll
  for (int i = 0; i < PPW_S; ++i) {                // V-tile image with QT column tiles: sub-tile (key slot half, column tile), 8 key rows x 64 bytes
    const int o = (wave * PPW_S + i) * 1024 + lane * 16;
    const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
    const int ct = sub % QT, sh = sub / QT;
    const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
    // workspace layout (tfa_bwd_kv_kernel.h): [key block of 128][query tile of 64][128 keys][64 queries]; this lane's 8 queries are
    // queries qo .. qo+7 of the workgroup's 256 = query tile q0/64 + qo/64, columns qo % 64
    const int qo = (ct * 4 + pcs) * 8;
    s_src[i] = ((((q0 >> 6) + (qo >> 6)) * 128 + key) * 64 + (qo & 63)) * 2;
  }
#pragma unroll
  for (int i = 0; i < PPW_K; ++i) {
    const int o = (wave * PPW_K + i) * 1024 + lane * 16;
    const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
    const int dt = sub % DT, sh = sub / DT;
    const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
    k_src[i] = (dt * 4 + pcs) * 8 < p.dv ? key * (int)p.k.s_n * 2 + ((dt * 4 + pcs) << 4) : (int)TFA_OOB;
  }
  const int k_tile_stride = BN * (int)p.k.s_n * 2;
  const int s_kb_stride = (p.ws_nq >> 6) * 128 * 64 * 2;              // bytes between the key blocks of a head's slab
  auto dma_issue = [&](int j, int stage) {
    const int s_off = (j >> 1) * s_kb_stride + (j & 1) * (64 * 64 * 2);   // key tile j = half (j & 1) of key block j / 2
#pragma unroll
    for (int i = 0; i < PPW_S; ++i)
      lds_dma16_m0_nt(s_rs, lds_base + stage * STAGE_BYTES + (wave * PPW_S + i) * 1024, s_src[i] + s_off);
#pragma unroll
    for (int i = 0; i < PPW_K; ++i)
      lds_dma16_m0(k_rs, lds_base + stage * STAGE_BYTES + S_BYTES + (wave * PPW_K + i) * 1024, k_src[i] + j * k_tile_stride);
  };

  f32x16 acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int rd_lo = ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const int s_rd_base = (hi * QT << 9) + rd_lo + (wave << 9);        // this wave's column tile of the dS^T image
  const int k_rd_base = (hi * DT << 9) + rd_lo;

  if (nt > 0) dma_issue(0, 0);
  if (nt > 1) dma_issue(1, 1);
  // tile 0 has landed when at most one tile's pieces are still in flight
  if (nt > 1) {
    if constexpr (PPW_S + PPW_K == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_barrier" ::: "memory");

  int st = 0, st2 = 2;                              // stage of tile j, stage of tile j+2
#pragma nounroll
  for (int j = 0; j < nt; ++j) {
    if (j + 2 < nt) dma_issue(j + 2, st2);          // that stage held tile j-1: every wave left it at the last barrier
    const char* simg = smem + st * STAGE_BYTES;
    const char* kimg = simg + S_BYTES;
    const bool active = !CAUSAL || (j * BN <= wave_q0 + 31 + shift);
    if (active) {
      X8 y[4];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const char* a = simg + s_rd_base + (sl * 2 * QT << 9);
        const s16x4 lo = lds_read_tr16_b64(a);
        const s16x4 hh = lds_read_tr16_b64(a + 256);
        y[sl] = __builtin_bit_cast(X8, __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const char* a = kimg + k_rd_base + (sl * 2 * DT << 9) + (d << 9);
          const s16x4 lo = lds_read_tr16_b64(a);
          const s16x4 hh = lds_read_tr16_b64(a + 256);
          const s16x8 kf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[d] = E::mfma(__builtin_bit_cast(X8, kf), y[sl], acc[d]);
        }
    }
    // tile j+1 (issued one iteration ago) must have landed; tile j+2's pieces may stay in flight
    if (j + 2 < nt) {
      if constexpr (PPW_S + PPW_K == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    st = st + 1 == NSTAGE ? 0 : st + 1;
    st2 = st2 + 1 == NSTAGE ? 0 : st2 + 1;
  }

  // ---- epilogue: acc[dt][r] = dQ[row my_row][32*dt + (r&3) + 8*(r>>2) + 4*hi] / scale ---------------------------------
  const float osc = p.scale;
  if (F32OUT) {
    float* gb = reinterpret_cast<float*>(p.grad) + b * p.gs_b + h * p.gs_h;
    auto g_rs = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, p.g_bytes, 0x00020000);
    const int goff = my_row * (int)p.gs_n * 4 + hi * 16;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v4 = {acc[d][4 * g4 + 0] * osc, acc[d][4 * g4 + 1] * osc, acc[d][4 * g4 + 2] * osc, acc[d][4 * g4 + 3] * osc};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), g_rs, d * 32 + g4 * 8 + hi * 4 < p.dv ? goff + (d * 32 + g4 * 8) * 4 : (int)TFA_OOB, 0, 0);
      }
  } else {
    T* gb = reinterpret_cast<T*>(p.grad) + b * p.gs_b + h * p.gs_h;
    auto g_rs = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, p.g_bytes, 0x00020000);
    const int goff = my_row * (int)p.gs_n * 2 + hi * 8;
    typedef __attribute__((ext_vector_type(4))) T t4;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        t4 v4 = {(T)(acc[d][4 * g4 + 0] * osc), (T)(acc[d][4 * g4 + 1] * osc), (T)(acc[d][4 * g4 + 2] * osc), (T)(acc[d][4 * g4 + 3] * osc)};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v4), g_rs, d * 32 + g4 * 8 + hi * 4 < p.dv ? goff + (d * 32 + g4 * 8) * 2 : (int)TFA_OOB, 0, 0);
      }
  }
}

}  // namespace tfa
