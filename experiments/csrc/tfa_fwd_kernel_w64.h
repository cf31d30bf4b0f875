// tfa_fwd_kernel_w64.h — 4 waves x 64 query rows, ONE wave per SIMD with the whole 512-entry
// register file: O accumulators (128 registers per lane) are PINNED in the accumulator half (AGPRs)
// by issuing the O += P V MFMAs as inline asm with "+a" operands, and the Q fragments (64 registers)
// are read by the QK^T MFMAs straight out of AGPRs ("a" B-operand); S 64, P 32 and the K/V fragments
// live in the 256 architectural VGPRs.
//
// Why: with 32 rows per wave every wave re-reads the whole K and V tile from LDS (256 KiB of
// fragment reads per 256x64 tile per CU = half the LDS read peak at MFMA peak, measured 38 % LDS
// busy) and the two waves of a SIMD fight over the matrix pipe.  With 64 rows per wave each K/V
// fragment read feeds TWO MFMAs (LDS traffic per flop halves), the barrier joins 4 waves, and the
// two 32-row blocks of a wave are independent instruction streams: softmax of block 0 sits beside
// the QK^T MFMAs of block 1, softmax of block 1 beside the PV MFMAs of block 0.
// Left to itself hipcc cannot place this kernel in 512 registers (RB=2 of tfa_fwd_kernel.h spills
// 107 VGPRs); pinning O in AGPRs takes the 128 hottest registers out of its hands.
//
// Inline-asm MFMA hazards handled here (nothing inside an asm string is padded by the compiler):
//   * VALU (cvt_pk) / v_accvgpr_write  -> MFMA operand: `s_nop 3` opens every MFMA string;
//   * MFMA result in AGPRs -> any non-MFMA reader (rescale, epilogue): `acc_fence()` (16 wait states)
//     tied to the accumulators precedes every such read.
#pragma once
#include "tfa_fwd_kernel_dma.h"

namespace tfa {

template <typename T> struct MfmaAsm;
template <> struct MfmaAsm<__bf16> {
  // O += A.B with the accumulator in AGPRs
  static __device__ __forceinline__ void acc(f32x16& c, bf16x8 a, bf16x8 b) {
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
  // S = A.B (+S) with the accumulator in VGPRs and the B operand (Q fragment) resident in AGPRs
  static __device__ __forceinline__ void qk0(f32x16& c, bf16x8 a, bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %1, 0" : "=&v"(c), "+a"(b) : "v"(a));
  }
  static __device__ __forceinline__ void qk(f32x16& c, bf16x8 a, bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "+v"(c), "+a"(b) : "v"(a));
  }
};
template <> struct MfmaAsm<_Float16> {
  static __device__ __forceinline__ void acc(f32x16& c, f16x8 a, f16x8 b) {
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ void qk0(f32x16& c, f16x8 a, f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, 0" : "=&v"(c), "+a"(b) : "v"(a));
  }
  static __device__ __forceinline__ void qk(f32x16& c, f16x8 a, f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(c), "+a"(b) : "v"(a));
  }
};

// AB: timing-only ablation bits (1: no exponentials, 2: no barrier, 4: no DMA, 8: no max/mask) — results are wrong when set
template <typename T, int D, bool CAUSAL, bool F32OUT, int VF, int AB = 0>
__global__ __launch_bounds__(256, 1) void fwd_kernel_w64(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int NW = 4;
  constexpr int RB = 2;                            // 32-row blocks per wave
  constexpr int BM = NW * 32 * RB;                 // 256 query rows per workgroup
  constexpr int BN = 64;
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int NBUF = 3;
  constexpr int PD = 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NW;                 // 4 (D=128) or 2 (D=64)
  constexpr int DS = D / 16;
  constexpr int DT = D / 32;
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  static_assert(PPW == 2 || PPW == 4, "");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const kl = smem;
  char* const vl = smem + NBUF * TILE_BYTES;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, rt_start = 0;
  if (p.trace) { rt_start = __builtin_amdgcn_s_memrealtime(); t_start = __builtin_amdgcn_s_memtime(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31;
  const int hi = lane >> 5;
  const int shift = p.shift;

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
  auto dma_issue = [&](int j, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave * PPW + i;
      lds_dma16(k_rs, lds_base + buf * TILE_BYTES + pc * 1024, k_src[i] + j * k_tile_stride);
      lds_dma16(v_rs, lds_base + (NBUF + buf) * TILE_BYTES + pc * 1024, v_src[i] + j * v_tile_stride);
    }
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

    if (nt > 0) dma_issue(0, 0);
    if (nt > 1) dma_issue(1, 1);
    if (nt > 2) dma_issue(2, 2);
    const int wave_row0 = q0 + wave * (32 * RB);
    int my_row[RB];
    X8 qf[RB][DS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      my_row[rb] = wave_row0 + rb * 32 + qi;
      const int qoff = my_row[rb] * (int)p.qs_n * 2 + hi * 16;
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(q_rs, qoff + s * 32, 0, 0);
        qf[rb][s] = __builtin_bit_cast(X8, t);
      }
    }
    f32x16 oacc[RB][DT];                           // pinned in AGPRs by the "+a" operands below
    float m_run[RB], l_run[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[rb][d][r] = 0.f;
      m_run[rb] = -1e30f;
      l_run[rb] = 0.f;
    }
    // every MFMA result must have left the pipe before a non-MFMA instruction touches the accumulators
    auto acc_fence = [&]() {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int d = 0; d < DT; ++d) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(oacc[rb][d]));
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[rb][s]));
    asm volatile("s_barrier" ::: "memory");
    if (p.trace && pass == 0) t_pro = __builtin_amdgcn_s_memtime();

    const int wave_last_tile = CAUSAL ? ((wave_row0 + 32 * RB - 1 + shift) >= 0 ? (wave_row0 + 32 * RB - 1 + shift) / BN : -1) : (nt - 1);

    // K fragments of the tile about to be computed (loop carried: read during the previous tile's stage D)
    X8 kf[DS][2];
    auto read_k = [&](int buf) {
      const char* kb = kl + buf * TILE_BYTES;
#pragma unroll
      for (int s = 0; s < DS; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int off = k_rd_base + t * 32 * (D * 2) + (((2 * s + hi) ^ k_rd_swz) << 4);
          kf[s][t] = __builtin_bit_cast(X8, lds_read_b128(kb, off));
        }
    };
    if (0 <= wave_last_tile && nt > 0) read_k(0);

    // Tile j (buffer j%3):  A: S0=K Q0^T | V reads | max0 | B: S1=K Q1^T + exp(0) | max1 | C: O0+=P0 V + exp(1)
    //   | wait tile j+1, BARRIER (every wave is done reading tile j) | DMA tile j+3 -> buffer j%3 |
    //   K reads of tile j+1 | D: O1+=P1 V
    auto tile_body = [&](int j, int buf) {
      const bool active = (j <= wave_last_tile);
      const char* vb = vl + buf * TILE_BYTES;
      const int key0 = j * BN;
      f32x16 sacc[RB][2];
      X8 pk[RB][4];
      s16x8 vfr[DT][4];

      auto mask_max = [&](int rb, f32x16 (&sc2)[2]) -> float {
        bool need_mask = (key0 + BN > p.Nk);
        if (CAUSAL) need_mask = need_mask || (key0 + BN - 1 > wave_row0 + rb * 32 + shift);
        if (need_mask) {
          int lim = p.Nk - 1;
          if (CAUSAL) { const int c = my_row[rb] + shift; lim = c < lim ? c : lim; }
          lim -= key0 + 4 * hi;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ko = 32 * t + (r & 3) + 8 * (r >> 2);
              if (ko > lim) sc2[t][r] = -INFINITY;
            }
        }
        float mloc = sc2[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sc2[t][r]);
        mloc = pair_max(mloc);
        return fmaxf(m_run[rb], mloc);
      };
      auto rescale = [&](int rb, float m_new) {
        if (__any(m_new != m_run[rb])) {
#pragma unroll
          for (int d = 0; d < DT; ++d) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(oacc[rb][d]));
          const float alpha = fast_exp2((m_run[rb] - m_new) * sc);
          l_run[rb] *= alpha;
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[rb][d][r] *= alpha;
        }
        m_run[rb] = m_new;
      };
      // two exponentials of a row block (elements 2*idx, 2*idx+1 of its 32) -> row sums, 16-bit P.
      // The empty asm statements pin each piece between the MFMAs around it.
      auto exp2_pair = [&](const f32x16 (&sc2)[2], float msc, int idx, float (&lsum)[4], X8 (&pkk)[4]) {
        const int e = idx * 2, t = e >> 4, r = e & 15;      // r is even: both elements land in one 32-bit P word
        float e0, e1;
        if (AB & 1) { e0 = sc2[t][r]; e1 = sc2[t][r + 1]; }
        else {
          e0 = fast_exp2(fmaf(sc2[t][r], sc, -msc));
          e1 = fast_exp2(fmaf(sc2[t][r + 1], sc, -msc));
          lsum[r & 3] += e0;
          lsum[(r + 1) & 3] += e1;
        }
        pkk[t * 2 + (r >> 3)][r & 7] = (T)e0;
        pkk[t * 2 + (r >> 3)][(r + 1) & 7] = (T)e1;
        asm volatile("" : "+v"(lsum[r & 3]), "+v"(lsum[(r + 1) & 3]), "+v"(pkk[t * 2 + (r >> 3)]));
      };

      if (active) {
        // ---- stage A: S0 = K Q0^T (matrix only) ---------------------------------------------------
#pragma unroll
        for (int s = 0; s < DS; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (s == 0) MfmaAsm<T>::qk0(sacc[0][t], kf[s][t], qf[0][s]);
            else MfmaAsm<T>::qk(sacc[0][t], kf[s][t], qf[0][s]);
          }
        // V fragments of this tile: issued now, consumed from stage C on
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
            s16x4 lo = lds_read_tr16_b64(a);
            s16x4 hh = lds_read_tr16_b64(a + 256);
            vfr[d][s] = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sacc[0][t]));
        if (!(AB & 8)) rescale(0, mask_max(0, sacc[0]));
        const float msc0 = m_run[0] * sc;

        // ---- stage B: S1 = K Q1^T, exponentials of row block 0 behind every MFMA ------------------
        float lsum0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DS; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (s == 0) MfmaAsm<T>::qk0(sacc[1][t], kf[s][t], qf[1][s]);
            else MfmaAsm<T>::qk(sacc[1][t], kf[s][t], qf[1][s]);
            constexpr int PER = 16 / (DS * 2);
#pragma unroll
            for (int q = 0; q < PER; ++q) exp2_pair(sacc[0], msc0, (s * 2 + t) * PER + q, lsum0, pk[0]);
          }
        l_run[0] += (lsum0[0] + lsum0[1]) + (lsum0[2] + lsum0[3]);
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sacc[1][t]));
        if (!(AB & 8)) rescale(1, mask_max(1, sacc[1]));
        const float msc1 = m_run[1] * sc;

        // ---- stage C: O0 += P0 V, exponentials of row block 1 behind every MFMA --------------------
        float lsum1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            MfmaAsm<T>::acc(oacc[0][d], __builtin_bit_cast(X8, vfr[d][s]), pk[0][s]);
            constexpr int PER = 16 / (4 * DT);
#pragma unroll
            for (int q = 0; q < PER; ++q) exp2_pair(sacc[1], msc1, (s * DT + d) * PER + q, lsum1, pk[1]);
          }
        l_run[1] += (lsum1[0] + lsum1[1]) + (lsum1[2] + lsum1[3]);
      }

      // ---- every LDS read of tile j has returned; tile j+1 must have landed; buffer j%3 becomes free ----
      if (j + 2 < nt) {
        if (PPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile j+2 may stay in flight
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (AB & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (j + 3 < nt && !(AB & 4)) dma_issue(j + 3, buf);
      if (j + 1 < nt && j + 1 <= wave_last_tile) read_k((buf + 1) % NBUF);

      if (active) {
        // ---- stage D: O1 += P1 V (matrix only; the K reads of the next tile land underneath) -------
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int d = 0; d < DT; ++d) MfmaAsm<T>::acc(oacc[1][d], __builtin_bit_cast(X8, vfr[d][s]), pk[1][s]);
      }
    };

    for (int j = 0; j < nt; j += 3) {
      tile_body(j, 0);
      if (j + 1 < nt) tile_body(j + 1, 1);
      if (j + 2 < nt) tile_body(j + 2, 2);
    }
    if (p.trace && pass == 0) t_loop = __builtin_amdgcn_s_memtime();

    // ---- epilogue ---------------------------------------------------------------------------
    acc_fence();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float l_tot = pair_sum(l_run[rb]);
      const bool empty = !(l_tot > 0.f);
      const float inv = empty ? 1.f : 1.f / l_tot;
      if (p.lse != nullptr && hi == 0 && my_row[rb] < p.Nq) {
        const float lse = empty ? INFINITY : (m_run[rb] * p.scale + __builtin_amdgcn_logf(l_tot) * 0.6931471805599453f);
        p.lse[(long long)bh * p.Nq + my_row[rb]] = lse;
      }
      if (F32OUT) {
        float* obase = reinterpret_cast<float*>(p.o) + b * p.os_b + h * p.os_h;
        auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
        const int ooff = my_row[rb] * (int)p.os_n * 4 + hi * 16;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v4 = {oacc[rb][d][4 * g + 0] * inv, oacc[rb][d][4 * g + 1] * inv, oacc[rb][d][4 * g + 2] * inv, oacc[rb][d][4 * g + 3] * inv};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), o_rs, ooff + (d * 32 + g * 8) * 4, 0, 0);
          }
      } else {
        T* obase = reinterpret_cast<T*>(p.o) + b * p.os_b + h * p.os_h;
        auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
        const int ooff = my_row[rb] * (int)p.os_n * 2 + hi * 8;
        typedef __attribute__((ext_vector_type(4))) T t4;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            t4 v4 = {(T)(oacc[rb][d][4 * g + 0] * inv), (T)(oacc[rb][d][4 * g + 1] * inv), (T)(oacc[rb][d][4 * g + 2] * inv), (T)(oacc[rb][d][4 * g + 3] * inv)};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v4), o_rs, ooff + (d * 32 + g * 8) * 2, 0, 0);
          }
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
