// tfa_fwd_kernel_pp.h — "ping-pong" schedule of the FlashAttention-2 forward tile loop (gfx950).
//
// Same math, layouts and fragments as tfa_fwd_kernel.h; what changes is WHEN each wave does what.
// A 512-thread workgroup puts two waves on every SIMD (wave w and wave w+4).  If both run the tile
// loop in lock-step their MFMA phases collide on the SIMD's one matrix pipe and their softmax phases
// collide on its VALU, and nothing overlaps.  Here the waves form two groups, g0 = waves 0-3 and
// g1 = waves 4-7, running the SAME instruction stream half a KV tile apart:
//
//      slot:      2j              2j+1             2j+2             2j+3
//      g0:   first half(j)   second half(j)   first half(j+1)  second half(j+1)
//      g1:   second half(j-1) first half(j)   second half(j)   first half(j+1)
//
//   first half  = K fragments LDS->VGPR, S^T = K Q^T (16 MFMA), V fragment reads issued, row max
//   second half = exp2 / row sum / 16-bit P (VALU, transcendental), O^T += V^T P^T (16 MFMA)
//
// so one group's matrix work always sits beside the other group's VALU/LDS work.  The stagger is held
// by the barriers themselves: every tile has a middle barrier M and an end barrier T; g1 executes one
// extra barrier before its loop and g0 one extra after, so g0's M_j is g1's T_{j-1}.
//
// LDS hand-off (two K and two V buffers, tile j in buffer j&1):
//   tile j+1 is written (by all 512 threads) in slot 2j+1: g0 right after M_j, g1 at the top of its
//   first half; it is first read in slot 2j+2.  The buffer it overwrites held tile j-1, last read in
//   slot 2j.  Global loads for tile j+2 are issued right after the LDS write of tile j+1 (the staging
//   registers are the third pipeline stage), so HBM/L2 latency has a whole tile to hide in.
#pragma once
#include "tfa_fwd_kernel.h"

namespace tfa {

// Issue-port padding (tools/probe_issue.hip): a wave streaming back-to-back MFMAs re-arms the SIMD's VALU issue port
// the moment it frees, so VALU work of the OTHER wave on that SIMD never gets in; an s_nop behind each MFMA leaves the gap.
// (VF_NOPQK_SHIFT / VF_NOPPV_SHIFT, tfa_fwd_kernel.h: "s_nop n-1" after every QK^T / PV MFMA)
template <int N, typename ACC> static __device__ __forceinline__ void issue_gap(ACC& acc) {
  if constexpr (N > 0) asm volatile("s_nop %1" : "+v"(acc) : "n"(N - 1));
}

// Barriers are inline asm (a memory clobber keeps LDS accesses on their side) followed by a full
// scheduling barrier (keeps register-only VALU/MFMA work of the next half from being hoisted above).
static __device__ __forceinline__ void barrier_raw() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
static __device__ __forceinline__ void barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int D, bool CAUSAL, bool F32OUT, int VF>
__global__ __launch_bounds__(512, 2) void fwd_kernel_pp(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int NW = 8;
  constexpr int BM = NW * 32;       // query rows per workgroup
  constexpr int BN = 64;            // keys per tile
  constexpr int NT = NW * 64;       // threads
  constexpr int CPR = D / 8;        // 16-byte chunks per row
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int NCH = BN * CPR / NT;  // staging chunks per thread per tensor
  constexpr int DS = D / 16;        // k-slots of the QK^T contraction
  constexpr int DT = D / 32;        // 32-wide d tiles of O
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  constexpr int VPRE = ((VF >> VF_VPRE_SHIFT) & 7) < DT ? ((VF >> VF_VPRE_SHIFT) & 7) : DT;
  static_assert(NCH >= 1, "tile too small for the workgroup");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const kl = smem;                    // K buffers 0,1
  char* const vl = smem + 2 * TILE_BYTES;   // V buffers 0,1

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, rt_start = 0;
  if (p.trace) { rt_start = __builtin_amdgcn_s_memrealtime(); t_start = __builtin_amdgcn_s_memtime(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                // 0: leads, 1: trails by half a tile
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

  int st_koff[NCH], st_voff[NCH], st_klds[NCH], st_vlds[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NT;
    const int row = c / CPR, cc = c % CPR;
    st_koff[i] = row * (int)p.ks_n * 2 + cc * 16;
    st_voff[i] = row * (int)p.vs_n * 2 + cc * 16;
    st_klds[i] = k_lds_off<D>(row, cc);
    st_vlds[i] = v_lds_off<D>(row, cc);
  }
  const int k_tile_stride = BN * (int)p.ks_n * 2;
  const int v_tile_stride = BN * (int)p.vs_n * 2;

  u32x4 kst[NCH], vst[NCH];
  auto stage_load = [&](int j) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      kst[i] = __builtin_amdgcn_raw_buffer_load_b128(k_rs, st_koff[i] + j * k_tile_stride, 0, 0);
      vst[i] = __builtin_amdgcn_raw_buffer_load_b128(v_rs, st_voff[i] + j * v_tile_stride, 0, 0);
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      lds_write_b128(kl, buf * TILE_BYTES + st_klds[i], kst[i]);
      lds_write_b128(vl, buf * TILE_BYTES + st_vlds[i], vst[i]);
    }
    asm volatile("" ::: "memory");   // keep these LDS writes ahead of the fragment reads that follow
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

    // ---- prologue ---------------------------------------------------------------------------
    const int my_row = q0 + wave * 32 + qi;
    X8 qf[DS];
    {
      const int qoff = my_row * (int)p.qs_n * 2 + hi * 16;
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(q_rs, qoff + s * 32, 0, 0);
        qf[s] = __builtin_bit_cast(X8, t);
      }
    }
    if (nt > 0) stage_load(0);

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -1e30f;
    float l_run = 0.f;

    if (nt > 0) stage_write(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));
    if (nt > 1) stage_load(1);
    barrier_lds();                              // P: tile 0 is in LDS
    if (p.trace && pass == 0) t_pro = __builtin_amdgcn_s_memtime();
    if (grp == 1) barrier_raw();                // X: g1 starts half a tile late

    const int wave_row0 = q0 + wave * 32;
    const int wave_last_tile = CAUSAL ? ((wave_row0 + 31 + shift) >= 0 ? (wave_row0 + 31 + shift) / BN : -1) : (nt - 1);

    auto tile_body = [&](int j, int buf) {
      const bool active = (j <= wave_last_tile);
      const char* kb = kl + buf * TILE_BYTES;
      const char* vb = vl + buf * TILE_BYTES;

      // ======================= first half =====================================================
      if (grp == 1) {
        if (j + 1 < nt) stage_write(buf ^ 1);
        if (j + 2 < nt) stage_load(j + 2);
      }
      f32x16 sacc[2];
      s16x8 vfr[VPRE > 0 ? VPRE : 1][4];
      float mloc = -INFINITY;
      if (active) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
        X8 kf[DS][2];
#pragma unroll
        for (int s = 0; s < DS; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int off = k_rd_base + t * 32 * (D * 2) + (((2 * s + hi) ^ k_rd_swz) << 4);
            kf[s][t] = __builtin_bit_cast(X8, lds_read_b128(kb, off));
          }
#pragma unroll
        for (int s = 0; s < DS; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            sacc[t] = E::mfma(kf[s][t], qf[s], sacc[t]);
            issue_gap<(VF >> VF_NOPQK_SHIFT) & 31>(sacc[t]);
          }

#pragma unroll
        for (int d = 0; d < VPRE; ++d)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
            s16x4 lo = lds_read_tr16_b64(a);
            s16x4 hh = lds_read_tr16_b64(a + 256);
            vfr[d][s] = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
          }

        const int key0 = j * BN;
        bool need_mask = (key0 + BN > p.Nk);
        if (CAUSAL) need_mask = need_mask || (key0 + BN - 1 > wave_row0 + shift);
        if (need_mask) {
          int lim = p.Nk - 1;
          if (CAUSAL) { const int c = my_row + shift; lim = c < lim ? c : lim; }
          lim -= key0 + 4 * hi;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ko = 32 * t + (r & 3) + 8 * (r >> 2);
              if (ko > lim) sacc[t][r] = -INFINITY;
            }
        }
        mloc = sacc[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[t][r]);
        mloc = pair_max(mloc);
      }

      barrier_raw();                            // M_j
      // ======================= second half ====================================================
      if (grp == 0) {
        if (j + 1 < nt) stage_write(buf ^ 1);
        if (j + 2 < nt) stage_load(j + 2);
      }
      if (active) {
        const float m_new = fmaxf(m_run, mloc);
        const bool changed = (m_new != m_run);
        if (__any(changed)) {
          const float alpha = fast_exp2((m_run - m_new) * sc);
          l_run *= alpha;
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        m_run = m_new;
        const float msc = m_new * sc;
        X8 pk[4];
        float lsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = fast_exp2(fmaf(sacc[t][r], sc, -msc));
            lsum[r & 3] += e;
            pk[t * 2 + (r >> 3)][r & 7] = (T)e;
          }
        l_run += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);

#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            s16x8 vf;
            if (d < VPRE) {
              vf = vfr[d][s];
            } else {
              const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
              s16x4 lo = lds_read_tr16_b64(a);
              s16x4 hh = lds_read_tr16_b64(a + 256);
              vf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            oacc[d] = E::mfma(__builtin_bit_cast(X8, vf), pk[s], oacc[d]);
            issue_gap<(VF >> VF_NOPPV_SHIFT) & 31>(oacc[d]);
          }
        }
      }
      barrier_lds();                            // T_j
    };

    for (int j = 0; j < nt; j += 2) {
      tile_body(j, 0);
      if (j + 1 < nt) tile_body(j + 1, 1);
    }
    if (grp == 0) barrier_raw();                // X': g0 waits for g1's last second half
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
