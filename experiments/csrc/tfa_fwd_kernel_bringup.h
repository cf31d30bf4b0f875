// tfa_fwd_kernel_bringup.h — the FIRST correct forward kernel of this repository (round 1): K/V tiles staged global -> registers ->
// LDS by the workgroup, burst-structured tile loop.  Kept for the A/B record only (variants 0-7 of tfa_launch.h, built with
// `make EXPERIMENTAL=1`); the product library never compiles it.  Layout helpers and KArgs: csrc/tfa_fwd_kernel.h.
#pragma once
#include "tfa_fwd_kernel.h"

namespace tfa {

// NW waves per workgroup; every wave owns RB blocks of 32 query rows (RB = 2: each K/V fragment
// read from LDS feeds two MFMAs, one wave per SIMD with the whole 512-entry register file).
template <typename T, int D, int NW, bool CAUSAL, bool F32OUT, int VF, int AB = 0, int RB = 1>
__global__ __launch_bounds__(NW * 64, RB == 1 ? 2 : 1) void fwd_kernel(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int WROWS = 32 * RB;    // query rows per wave
  constexpr int BM = NW * WROWS;    // query rows per workgroup
  constexpr int BN = 64;            // keys per tile
  constexpr int NT = NW * 64;       // threads
  constexpr int CPR = D / 8;        // 16-byte chunks per row
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int NCH = BN * CPR / NT;  // staging chunks per thread per tensor
  constexpr int DS = D / 16;        // k-slots of the QK^T contraction
  constexpr int DT = D / 32;        // 32-wide d tiles of O
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  constexpr bool KPRE = (VF & VF_KPRE) != 0;
  constexpr bool VPRE = (VF & VF_VPRE) && (VF & VF_TRREAD);
  static_assert(NCH >= 1, "tile too small for the workgroup");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const kl = smem;                    // K buffers 0,1
  char* const vl = smem + 2 * TILE_BYTES;   // V buffers 0,1

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, rt_start = 0;
  if (p.trace) { rt_start = __builtin_amdgcn_s_memrealtime(); t_start = __builtin_amdgcn_s_memtime(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31;
  const int hi = lane >> 5;

  // ---- workgroup -> (b,h, work item): heads of one XCD stay together, heavy blocks first
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
  const int shift = p.shift;           // causal: key j visible to row i iff j <= i + shift

  const T* qbase = reinterpret_cast<const T*>(p.q) + b * p.qs_b + h * p.qs_h;
  const T* kbase = reinterpret_cast<const T*>(p.k) + b * p.ks_b + hk * p.ks_h;
  const T* vbase = reinterpret_cast<const T*>(p.v) + b * p.vs_b + hk * p.vs_h;
  auto q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, (unsigned)p.q_bytes, 0x00020000);
  auto k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (unsigned)p.k_bytes, 0x00020000);
  auto v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (unsigned)p.v_bytes, 0x00020000);

  // ---- staging geometry (constant per thread) ----------------------------------------------
  int st_koff[NCH], st_voff[NCH];            // byte offset inside the (b,h) slice for tile 0
  int st_klds[NCH], st_vlds[NCH];            // LDS byte offsets inside one tile buffer
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
  };

  // per-lane LDS read bases
  const int k_rd_base = qi * (D * 2);                       // key row (lane&31) of key-tile 0
  const int k_rd_swz = k_swz<D>(qi);
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const int v_ga_base = (hi * DT << 9) + (qi << 1);         // gather variant: column (lane&31)
  const float sc = p.scale_log2;
  int nt_total = 0;

  // A paired (causal) work item walks the heavy block nmb-1-wi first, then the light block wi.
  const int npass = PAIR ? ((p.nmb - 1 - wi) != wi ? 2 : 1) : 1;
#pragma nounroll
  for (int pass = 0; pass < npass; ++pass) {
    int mb;
    if (PAIR) mb = pass == 0 ? (p.nmb - 1 - wi) : wi;
    else mb = CAUSAL ? (p.nmb - 1 - wi) : wi;
    const int q0 = mb * BM;

    // number of KV tiles this block walks
    int kv_end = p.Nk;
    if (CAUSAL) {
      const int lim = q0 + BM + shift;       // one past the last key any row of the block sees
      kv_end = lim < kv_end ? lim : kv_end;
    }
    const int nt = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
    nt_total += nt;

    // ---- prologue: Q fragments first (so no Q load is ever pending inside the tile loop),
    //      then the first K/V tile
    const int wave_row0 = q0 + wave * WROWS;
    int my_row[RB];                            // the query rows this lane owns
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
    if (nt > 0) stage_load(0);

    f32x16 oacc[RB][DT];
    float m_run[RB], l_run[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[rb][d][r] = 0.f;
      m_run[rb] = -1e30f;   // running max of the raw (unscaled) scores
      l_run[rb] = 0.f;      // this lane's partial row sum (its 32 keys per tile)
    }

    if (nt > 0) stage_write(0);
    // every load above has landed (the LDS write consumed the last one): pin that fact so the
    // compiler does not carry "Q may be pending" into the loop and drain vmcnt there
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[rb][s]));
    __syncthreads();
    if (p.trace && pass == 0) t_pro = __builtin_amdgcn_s_memtime();

    // last tile this wave needs (causal): rows [wave_row0, wave_row0 + WROWS)
    const int wave_last_tile = CAUSAL ? ((wave_row0 + WROWS - 1 + shift) >= 0 ? (wave_row0 + WROWS - 1 + shift) / BN : -1) : (nt - 1);

    auto tile_body = [&](int j, int buf) {
      const bool has_next = (j + 1 < nt) && !(AB & AB_NOSTAGE);
      if (has_next) stage_load(j + 1);

      if (j <= wave_last_tile) {
        const char* kb = kl + buf * TILE_BYTES;
        const char* vb = vl + buf * TILE_BYTES;

        // ---- S^T = K Q^T : two 32-key tiles x DS k-slots, RB query blocks ------------------
        f32x16 sacc[RB][2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[rb][t][r] = 0.f;
        if (KPRE) {
          X8 kf[DS][2];
#pragma unroll
          for (int s = 0; s < DS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int off = k_rd_base + t * 32 * (D * 2) + (((2 * s + hi) ^ k_rd_swz) << 4);
              if (AB & AB_NOKREAD) kf[s][t] = qf[0][(s + t) % DS];
              else kf[s][t] = __builtin_bit_cast(X8, lds_read_b128(kb, off));
            }
#pragma unroll
          for (int s = 0; s < DS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int rb = 0; rb < RB; ++rb) {
                if (AB & AB_NOQK) asm volatile("" ::"v"(kf[s][t]));
                else sacc[rb][t] = E::mfma(kf[s][t], qf[rb][s], sacc[rb][t]);
              }
        } else {
#pragma unroll
          for (int s = 0; s < DS; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int off = k_rd_base + t * 32 * (D * 2) + (((2 * s + hi) ^ k_rd_swz) << 4);
              X8 kf = __builtin_bit_cast(X8, lds_read_b128(kb, off));
#pragma unroll
              for (int rb = 0; rb < RB; ++rb) sacc[rb][t] = E::mfma(kf, qf[rb][s], sacc[rb][t]);
            }
          }
        }

        // ---- V fragments: issue every transpose read now, consume after the softmax -------
        s16x8 vfr[DT][4];
        if (VPRE) {
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
              if (AB & AB_NOVREAD) {
                vfr[d][s] = __builtin_bit_cast(s16x8, qf[0][(d + s) % DS]);
              } else {
                s16x4 lo = lds_read_tr16_b64(a);
                s16x4 hh = lds_read_tr16_b64(a + 256);
                vfr[d][s] = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
              }
            }
        }

        X8 pk[RB][4];
        const int key0 = j * BN;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          // ---- masking (causal diagonal / ragged last tile) -------------------------------
          bool need_mask = (key0 + BN > p.Nk);
          if (CAUSAL) need_mask = need_mask || (key0 + BN - 1 > wave_row0 + rb * 32 + shift);
          if (need_mask) {
            int lim = p.Nk - 1;                              // last valid key
            if (CAUSAL) { const int c = my_row[rb] + shift; lim = c < lim ? c : lim; }
            lim -= key0 + 4 * hi;                            // compare against the in-tile key offset
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int ko = 32 * t + (r & 3) + 8 * (r >> 2);
                if (ko > lim) sacc[rb][t][r] = -INFINITY;
              }
          }

          // ---- online softmax (row = lane&31; the two half-waves hold different keys) -----
          float mloc = sacc[rb][0][0];
          if (!(AB & AB_NOSM)) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[rb][t][r]);
            mloc = pair_max(mloc);
          }
          const float m_new = (AB & AB_NOSM) ? m_run[rb] : fmaxf(m_run[rb], mloc);
          const bool changed = (m_new != m_run[rb]);
          if ((VF & VF_NOSKIP) || __any(changed)) {
            const float alpha = fast_exp2((m_run[rb] - m_new) * sc);
            l_run[rb] *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
              for (int r = 0; r < 16; ++r) oacc[rb][d][r] *= alpha;
          }
          m_run[rb] = m_new;
          const float msc = m_new * sc;
          float lsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float e;
              if (AB & AB_NOSM) e = sacc[rb][t][r];
              else if (AB & AB_NOEXP) e = fmaf(sacc[rb][t][r], sc, -msc);
              else e = fast_exp2(fmaf(sacc[rb][t][r], sc, -msc));
              if (!(AB & AB_NOSM)) lsum[r & 3] += e;
              if (AB & AB_NOCVT) pk[rb][t * 2 + (r >> 3)][r & 7] = __builtin_bit_cast(T, (unsigned short)(__builtin_bit_cast(unsigned, e) >> 16));
              else pk[rb][t * 2 + (r >> 3)][r & 7] = (T)e;
            }
          l_run[rb] += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
        }

        // ---- O^T += V^T P^T  (k-slot outer, d-tile inner: consecutive MFMAs hit different accumulators)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            s16x8 vf;
            if (VPRE) {
              vf = vfr[d][s];
            } else if (VF & VF_TRREAD) {
              const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
              s16x4 lo = lds_read_tr16_b64(a);
              s16x4 hh = lds_read_tr16_b64(a + 256);
              vf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
              const char* a = vb + v_ga_base + (s * 2 * DT << 9) + (d << 9);
#pragma unroll
              for (int e = 0; e < 8; ++e) vf[e] = *reinterpret_cast<const short*>(a + (e << 6));
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
              if (AB & AB_NOPV) { asm volatile("" ::"v"(vf), "v"(pk[rb][s])); }
              else oacc[rb][d] = E::mfma(__builtin_bit_cast(X8, vf), pk[rb][s], oacc[rb][d]);
            }
          }
        }
      }

      if (has_next) stage_write(buf ^ 1);
      __syncthreads();
    };

    for (int j = 0; j < nt; j += 2) {
      tile_body(j, 0);
      if (j + 1 < nt) tile_body(j + 1, 1);
    }
    if (p.trace && pass == 0) t_loop = __builtin_amdgcn_s_memtime();

    // ---- epilogue: normalise, LSE, store ----------------------------------------------------
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float l_tot = pair_sum(l_run[rb]);
      const bool empty = !(l_tot > 0.f);                       // l == 0 or NaN  (flash_attention.cu:620)
      const float inv = empty ? 1.f : 1.f / l_tot;

      if (p.lse != nullptr && hi == 0 && my_row[rb] < p.Nq) {
        // LSE = m*scale + ln(l)  (flash_attention.cu:623); ln via log2
        const float lse = empty ? INFINITY : (m_run[rb] * p.scale + __builtin_amdgcn_logf(l_tot) * 0.6931471805599453f);
        p.lse[(long long)bh * p.Nq + my_row[rb]] = lse;
      }

      // lane holds, for its row, d = 32*dt + 8*g + 4*hi + {0..3}  (g = r>>2)
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
    // stores above are still in flight: drain them so t_end includes the store tail
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
