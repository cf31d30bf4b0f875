// tfa_fwd_kernel_swp.h — LDS-DMA staged forward kernel with a SOFTWARE-PIPELINED tile loop.
//
// tfa_fwd_kernel_dma.h runs, per wave and per KV tile,  QK^T (MFMA) -> softmax (VALU) -> PV (MFMA)
// strictly in that order, so inside one wave matrix work and vector work never overlap, and the two
// waves of a SIMD, synchronised by the per-tile barrier, tend to want the same pipe at the same time.
// Here iteration j of the loop carries TWO independent instruction streams in ONE basic block:
//
//     exp2 / row-sum / 16-bit pack of tile j   (VALU, transcendental)   ||   S(j+1) = K(j+1) Q^T  (MFMA + LDS)
//     O += P(j) V(j)                           (MFMA + LDS)            ||   row max of S(j+1)    (VALU)
//
// S(j+1) lives in a second accumulator set (32 more VGPRs), so the scheduler is free to interleave the
// exponentials of tile j between the QK^T MFMAs of tile j+1.  K tiles therefore run one tile ahead of
// V tiles in LDS: iteration j reads K(j+1) and V(j).  Three K and three V buffers; at the top of
// iteration j the DMA for K(j+3) and V(j+2) is issued into the buffers K(j) and V(j-1) vacated in
// iteration j-1; the counted vmcnt at the end of the iteration leaves exactly that youngest group in
// flight.  Tiles that need masking (causal diagonal, ragged tail) and the first/last tiles of a wave
// take a slower, unfused path through the same helpers.
#pragma once
#include "tfa_fwd_kernel_dma.h"

namespace tfa {

template <int N> static __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N == 0 || N == 1 || N == 2 || N == 4 || N == 8, "unsupported count");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <typename T, int D, bool CAUSAL, bool F32OUT, int VF>
__global__ __launch_bounds__(512, 2) void fwd_kernel_swp(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int NW = 8;
  constexpr int BM = NW * 32;
  constexpr int BN = 64;
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int NBUF = 3;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NW;                 // DMA pieces per wave per tensor per tile (1 or 2)
  constexpr int DS = D / 16;
  constexpr int DT = D / 32;
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  static_assert(PPW == 1 || PPW == 2, "");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const kl = smem;                           // K buffers 0..2
  char* const vl = smem + NBUF * TILE_BYTES;       // V buffers 0..2
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
    if ((p.nbh & 7) == 0) {
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
  auto q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, (unsigned)p.q_bytes, 0x00020000);
  auto k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (unsigned)p.k_bytes, 0x00020000);
  auto v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (unsigned)p.v_bytes, 0x00020000);

  int k_src[PPW], v_src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave * PPW + i;
    {
      const int row = pc * (1024 / (D * 2)) + lane / CPR;
      const int cpos = lane % CPR;
      k_src[i] = row * (int)p.ks_n * 2 + ((cpos ^ k_swz<D>(row)) << 4);
    }
    {
      const int o = pc * 1024 + lane * 16;
      const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
      const int dt = sub % DT, sh = sub / DT;
      const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
      v_src[i] = key * (int)p.vs_n * 2 + ((dt * 4 + pcs) << 4);
    }
  }
  const int k_tile_stride = BN * (int)p.ks_n * 2;
  const int v_tile_stride = BN * (int)p.vs_n * 2;

  auto dma_k = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      lds_dma16(k_rs, lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i] + t * k_tile_stride);
  };
  auto dma_v = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      lds_dma16(v_rs, lds_base + (NBUF + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i] + t * v_tile_stride);
  };

  const int k_rd_base = qi * (D * 2);
  const int k_rd_swz = k_swz<D>(qi);
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const float sc = p.scale_log2;
  int nt_total = 0;

  const int npass = PAIR ? ((p.nmb - 1 - wi) != wi ? 2 : 1) : 1;
#pragma nounroll
  for (int pass = 0; pass < npass; ++pass) {
    int mb;
    if (PAIR) mb = pass == 0 ? (p.nmb - 1 - wi) : wi;
    else mb = CAUSAL ? (p.nmb - 1 - wi) : wi;
    const int q0 = mb * BM;
    int kv_end = p.Nk;
    if (CAUSAL) {
      const int lim = q0 + BM + shift;
      kv_end = lim < kv_end ? lim : kv_end;
    }
    const int nt = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
    nt_total += nt;

    const int wave_row0 = q0 + wave * 32;
    const int my_row = wave_row0 + qi;
    const int wave_last_tile = CAUSAL ? ((wave_row0 + 31 + shift) >= 0 ? (wave_row0 + 31 + shift) / BN : -1) : (nt - 1);

    // ---- helpers over one tile --------------------------------------------------------------
    X8 qf[DS];
    f32x16 oacc[DT];
    float m_run = -1e30f, l_run = 0.f;

    auto qk_tile = [&](int kbuf, f32x16 (&s)[2]) {
      const char* kb = kl + kbuf * TILE_BYTES;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int sl = 0; sl < DS; ++sl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int off = k_rd_base + t * 32 * (D * 2) + (((2 * sl + hi) ^ k_rd_swz) << 4);
          X8 kf = __builtin_bit_cast(X8, lds_read_b128(kb, off));
          s[t] = E::mfma(kf, qf[sl], s[t]);
        }
    };
    auto needs_mask = [&](int t) -> bool {
      const int key0 = t * BN;
      bool nm = (key0 + BN > p.Nk);
      if (CAUSAL) nm = nm || (key0 + BN - 1 > wave_row0 + shift);
      return nm;
    };
    auto apply_mask = [&](int t, f32x16 (&s)[2]) {
      int lim = p.Nk - 1;
      if (CAUSAL) { const int c = my_row + shift; lim = c < lim ? c : lim; }
      lim -= t * BN + 4 * hi;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * tt + (r & 3) + 8 * (r >> 2);
          if (ko > lim) s[tt][r] = -INFINITY;
        }
    };
    auto row_max = [&](const f32x16 (&s)[2]) -> float {
      float mloc = s[0][0];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[t][r]);
      return pair_max(mloc);
    };
    // running max update + (rare) rescale of O; returns m_new * sc
    auto update_max = [&](float mloc) -> float {
      const float m_new = fmaxf(m_run, mloc);
      if (__any(m_new != m_run)) {
        const float alpha = fast_exp2((m_run - m_new) * sc);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      }
      m_run = m_new;
      return m_new * sc;
    };
    auto exp_pack = [&](const f32x16 (&s)[2], float msc, X8 (&pk)[4]) {
      float lsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = fast_exp2(fmaf(s[t][r], sc, -msc));
          lsum[r & 3] += e;
          pk[t * 2 + (r >> 3)][r & 7] = (T)e;
        }
      l_run += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
    };
    auto pv_tile = [&](int vbuf, const X8 (&pk)[4]) {
      const char* vb = vl + vbuf * TILE_BYTES;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const char* a = vb + v_rd_base + (sl * 2 * DT << 9) + (d << 9);
          s16x4 lo = lds_read_tr16_b64(a);
          s16x4 hh = lds_read_tr16_b64(a + 256);
          s16x8 vf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[d] = E::mfma(__builtin_bit_cast(X8, vf), pk[sl], oacc[d]);
          if (d == DT - 1) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: K(0..2), V(0..1) by DMA, Q fragments, S(0) -----------------------------------
    if (nt > 0) dma_k(0, 0);
    if (nt > 0) dma_v(0, 0);
    if (nt > 1) dma_k(1, 1);
    if (nt > 1) dma_v(1, 1);
    if (nt > 2) dma_k(2, 2);
    {
      const int qoff = my_row * (int)p.qs_n * 2 + hi * 16;
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(q_rs, qoff + s * 32, 0, 0);
        qf[s] = __builtin_bit_cast(X8, t);
      }
    }
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

    wait_vmcnt<0>();
#pragma unroll
    for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));
    asm volatile("s_barrier" ::: "memory");
    if (p.trace && pass == 0) t_pro = __builtin_amdgcn_s_memtime();

    // tiles this wave computes: 0 .. nact-1 (causal: waves of a block stop at different tiles)
    const int nact = (wave_last_tile + 1 < nt) ? (wave_last_tile + 1) : nt;

    f32x16 sA[2], sB[2];
    float mA = -INFINITY, mB = -INFINITY;
    if (nact > 0) {
      qk_tile(0, sA);
      if (needs_mask(0)) apply_mask(0, sA);
      mA = row_max(sA);
    }
    // K buffer 0 is about to be refilled with K(3): every wave must be done with K(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int kb = 1, vb = 0;   // K buffer of tile j+1, V buffer of tile j
    // top of iteration j: DMA for K(j+3) -> K buffer j%3 == (kb+2)%3, V(j+2) -> V buffer (vb+2)%3
    // end of iteration j: K(j+2) and V(j+1) must have landed (only the group issued at the top may stay
    // in flight), every wave must be done reading K(j+1) and V(j); then the buffer indices advance.
    auto iter_begin = [&](int j) -> int {
      const bool issue_k = (j + 3 < nt), issue_v = (j + 2 < nt);
      if (issue_k) dma_k(j + 3, (kb + 2) % NBUF);
      if (issue_v) dma_v(j + 2, (vb + 2) % NBUF);
      return (issue_k ? PPW : 0) + (issue_v ? PPW : 0);
    };
    auto iter_end = [&](int young) {
      if (young == 2 * PPW) wait_vmcnt<2 * PPW>();
      else if (young == PPW) wait_vmcnt<PPW>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      kb = kb == 2 ? 0 : kb + 1;
      vb = vb == 2 ? 0 : vb + 1;
    };

    // ---- steady-state iteration: tile j from (scur, mcur); produces (snext, mnext) for tile j+1 ----
    // Written already interleaved so that fragment reads stay just-in-time (register pressure).
    auto fused = [&](int j, f32x16 (&scur)[2], float mcur, f32x16 (&snext)[2], float& mnext) {
      const int young = iter_begin(j);
      X8 pk[4];
      const float msc = update_max(mcur);
      const char* kbp = kl + kb * TILE_BYTES;
      const char* vbp = vl + vb * TILE_BYTES;
      // part 1, per k-slot: 2 K reads, 2 MFMA of S(j+1), 4 exponentials of tile j
      float lsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < DS; ++sl) {
        const int off0 = k_rd_base + (((2 * sl + hi) ^ k_rd_swz) << 4);
        X8 kf0 = __builtin_bit_cast(X8, lds_read_b128(kbp, off0));
        X8 kf1 = __builtin_bit_cast(X8, lds_read_b128(kbp, off0 + 32 * (D * 2)));
        if (sl == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          snext[0] = E::mfma(kf0, qf[sl], z);
          snext[1] = E::mfma(kf1, qf[sl], z);
        } else {
          snext[0] = E::mfma(kf0, qf[sl], snext[0]);
          snext[1] = E::mfma(kf1, qf[sl], snext[1]);
        }
        constexpr int EPS = 32 / DS;            // exponentials per k-slot
#pragma unroll
        for (int e0 = 0; e0 < EPS; ++e0) {
          const int e = sl * EPS + e0, t = e >> 4, r = e & 15;
          const float ev = fast_exp2(fmaf(scur[t][r], sc, -msc));
          lsum[r & 3] += ev;
          pk[t * 2 + (r >> 3)][r & 7] = (T)ev;
        }
      }
      l_run += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
      if (needs_mask(j + 1)) apply_mask(j + 1, snext);   // causal diagonal / ragged tail only
      // part 2, per MFMA of O += P(j) V(j): 2 transpose reads, 2 inputs of the row max of S(j+1)
      float mx = -INFINITY;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const char* a = vbp + v_rd_base + (sl * 2 * DT << 9) + (d << 9);
          s16x4 lo = lds_read_tr16_b64(a);
          s16x4 hh = lds_read_tr16_b64(a + 256);
          s16x8 vf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[d] = E::mfma(__builtin_bit_cast(X8, vf), pk[sl], oacc[d]);
          constexpr int MPS = 32 / (4 * DT);    // row-max inputs per MFMA
#pragma unroll
          for (int e0 = 0; e0 < MPS; ++e0) {
            const int e = (sl * DT + d) * MPS + e0;
            mx = fmaxf(mx, snext[e >> 4][e & 15]);
          }
        }
      mnext = pair_max(mx);
      iter_end(young);
    };

    int j = 0;
    const int nfused = nact > 0 ? nact - 1 : 0;          // iterations that also produce S(j+1)
    if (nfused & 1) {                                    // odd count: one iteration, then rename B -> A
      fused(0, sA, mA, sB, mB);
#pragma unroll
      for (int t = 0; t < 2; ++t) sA[t] = sB[t];
      mA = mB;
      j = 1;
    }
    for (; j < nfused; j += 2) {
      fused(j, sA, mA, sB, mB);
      fused(j + 1, sB, mB, sA, mA);
    }
    if (nact > 0) {                                      // the wave's last tile: nothing to prefetch
      const int young = iter_begin(j);
      X8 pk[4];
      const float msc = update_max(mA);
      exp_pack(sA, msc, pk);
      pv_tile(vb, pk);
      iter_end(young);
      ++j;
    }
    for (; j < nt; ++j) {                                // tiles of the block this wave does not touch
      const int young = iter_begin(j);
      iter_end(young);
    }
    if (p.trace && pass == 0) t_loop = __builtin_amdgcn_s_memtime();

    // ---- epilogue ---------------------------------------------------------------------------
    const float l_tot = pair_sum(l_run);
    const bool empty = !(l_tot > 0.f);
    const float inv = empty ? 1.f : 1.f / l_tot;
    if (p.lse != nullptr && hi == 0 && my_row < p.Nq) {
      const float lse = empty ? INFINITY : (m_run * p.scale + __builtin_amdgcn_logf(l_tot) * 0.6931471805599453f);
      p.lse[(long long)bh * p.Nq + my_row] = lse;
    }
    if (F32OUT) {
      float* obase = reinterpret_cast<float*>(p.o) + b * p.os_b + h * p.os_h;
      auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
      const int ooff = my_row * (int)p.os_n * 4 + hi * 16;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v4 = {oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), o_rs, ooff + (d * 32 + g * 8) * 4, 0, 0);
        }
    } else {
      T* obase = reinterpret_cast<T*>(p.o) + b * p.os_b + h * p.os_h;
      auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
      const int ooff = my_row * (int)p.os_n * 2 + hi * 8;
      typedef __attribute__((ext_vector_type(4))) T t4;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          t4 v4 = {(T)(oacc[d][4 * g + 0] * inv), (T)(oacc[d][4 * g + 1] * inv), (T)(oacc[d][4 * g + 2] * inv), (T)(oacc[d][4 * g + 3] * inv)};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v4), o_rs, ooff + (d * 32 + g * 8) * 2, 0, 0);
        }
    }
  }

  if (p.trace) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
      t[0] = t_start; t[1] = t_pro; t[2] = t_loop; t[3] = t_end;
      t[4] = (unsigned long long)nt_total;
      t[5] = (unsigned long long)__builtin_amdgcn_s_getreg(63508) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32);   // XCC_ID | HW_ID << 32
      t[6] = __builtin_amdgcn_s_memrealtime() - rt_start;   // 100 MHz ticks over the same span as t[3] - t[0] shader cycles
      t[7] = ((unsigned long long)bh << 32) | (unsigned)wi;
    }
  }
}

}  // namespace tfa
