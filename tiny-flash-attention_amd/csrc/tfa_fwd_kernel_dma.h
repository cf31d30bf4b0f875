// tfa_fwd_kernel_dma.h — the forward tile loop with K/V tiles brought into LDS by LDS-DMA
// (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction) instead of global->VGPR->ds_write.
//
// Why: in the register-staged kernel (tfa_fwd_kernel.h) the staging costs ~13 % of the tile time
// (ablation: tools/ablate.py, NOSTAGE): 4 ds_write_b128 per thread per tile occupy the LDS store
// path and the single register set limits the global prefetch distance to one tile.  Here
//   * no staging VGPRs and no ds_write: the DMA writes LDS directly;
//   * three LDS tile buffers: tile j is computed while tile j+1 is landed/landing and tile j+2 is
//     in flight — two tiles of latency tolerance, waits are COUNTED (`s_waitcnt vmcnt(N)`, never a
//     drain while a younger tile is in flight) and barriers are raw `s_barrier`;
//   * an LDS-DMA piece lands lane-linear (wave-uniform base + lane*16), so the K chunk swizzle and
//     the V sub-tile order are applied to the per-lane SOURCE address; the LDS images are exactly
//     those of tfa_fwd_kernel.h (same fragment reads).
// Out-of-range rows still read as zeros (buffer descriptor bounds check), so ragged N is unchanged.
#pragma once
#include "tfa_fwd_kernel.h"
#include "tfa_acc_regs.h"

namespace tfa {

// One LDS-DMA piece: every lane fetches 16 bytes at (descriptor base + voffset) and the wave's
// 1 KiB lands at LDS byte address lds_addr + lane*16 (M0 = wave-uniform LDS address).
// Inline asm on purpose: given the builtin, hipcc (ROCm 7.2) orders every later ds_read that may
// alias behind the DMA with `s_waitcnt vmcnt(0)`, which drains the two-tile-deep pipeline each
// tile.  Nothing here has a VGPR destination; completion is tracked by the counted vmcnt waits in
// the tile loop.  `s_nop 4` covers "SALU wrote an SGPR of the descriptor / M0 -> VMEM reads it".
static __device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voffset), "s"(rs)
      : "memory");
}

// lds_dma16 with the non-temporal hint: K/V bytes no other workgroup will ask for (decode: one query block per K/V head)
static __device__ __forceinline__ void lds_dma16_nt(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voffset) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %2, %3, 0 offen nt lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voffset), "s"(rs)
      : "memory");
}

// LDS-DMA without saving/restoring M0 around it (2 SALU less per piece in the hot loop).  Safe only because nothing else
// in the kernels that call it (fwd_kernel_il, bwd_kernel) uses M0 (hipcc emits no M0 user here: LDS instructions need none on gfx9+, SGPR spills use immediate
// lane indices); tests/test_abi.py disassembles the library and fails if that ever changes.
static __device__ __forceinline__ void lds_dma16_m0(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voffset) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"                                     // SALU write of M0 -> LDS-DMA reads it: 1 wait state
      "buffer_load_dwordx4 %1, %2, 0 offen lds"
      :
      : "s"(lds_addr), "v"(voffset), "s"(rs)
      : "memory", "m0");
}

// The 4-byte form: one dword per lane, 256 contiguous bytes of LDS per wave-instruction (the backward's per-tile row statistics)
static __device__ __forceinline__ void lds_dma4_m0(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voffset) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "buffer_load_dword %1, %2, 0 offen lds"
      :
      : "s"(lds_addr), "v"(voffset), "s"(rs)
      : "memory", "m0");
}

// The same with the non-temporal hint: K/V bytes nobody else will ask for (decode: one workgroup per K/V head)
static __device__ __forceinline__ void lds_dma16_m0_nt(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voffset) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen nt lds"
      :
      : "s"(lds_addr), "v"(voffset), "s"(rs)
      : "memory", "m0");
}

// LDS-DMA through a descriptor computed just before (per-tile windows): a VMEM instruction reading an SGPR that SALU code
// wrote needs 5 wait states, and nothing pads the inside of an asm.
static __device__ __forceinline__ void lds_dma16_m0_fresh(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voffset) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds"
      :
      : "s"(lds_addr), "v"(voffset), "s"(rs)
      : "memory", "m0");
}

constexpr int VF_PERSIST = 2048;   // launch one workgroup per CU and walk the work items (else one item per workgroup)
constexpr int VF_2BUF = 4096;      // two LDS tile buffers (64 KiB at D=128): two 4-wave workgroups fit one CU
constexpr int VF_LDSEPI = 16384;   // epilogue: transpose O through LDS and store whole rows (16-byte coalesced stores)
constexpr int VF_DMA_NT = 1 << 27;   // K/V tiles streamed with the non-temporal hint (decode over a K/V cache larger than the memory-side cache)
constexpr int VF_PRIO = 1024;   // s_setprio(1) around the MFMA clusters (experiment, tests/tools/ab.py)

// The kernel walks a STREAM of query blocks: workgroup g takes work items g, g+G, g+2G, ... (G =
// gridDim.x; a causal work item is the pair {heavy block nmb-1-i, light block i}, so every item costs
// the same).  With G = number of CUs the launch is persistent: no workgroup turn-around between
// blocks (measured ~5.2k cycles each), and the next block's first two K/V tiles and its Q fragments are
// requested BEFORE the current block's epilogue, so its prologue latency hides behind the O stores.
// With G = number of items the same code degenerates to one item per workgroup.
template <typename T, int D, int NW, bool CAUSAL, bool F32OUT, int VF, int AB = 0>
// D = 256 (WIDE): the partial pass of tfa_fwd_splitkv at head dims above 128 — a K/V streaming kernel for decode-like shapes.  One wave
// per SIMD with the whole register file; O lives in the hand-owned AccVGPRs a[0:127] and Q is read from AccVGPRs (tfa_acc_regs.h:
// left to hipcc's own AccVGPR plan this instantiation spills 252 registers and streams 8-9 GB/s per workgroup); LDS fragments are
// read a group ahead of the asm MFMAs.
__global__ __launch_bounds__(NW * 64, D > 128 ? 1 : 2) void fwd_kernel_dma(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int BM = NW * 32;
  constexpr int BN = 64;
  constexpr int CPR = D / 8;                       // 16-byte chunks per row
  constexpr int TILE_BYTES = BN * D * 2;           // one K (or V) tile
  constexpr int NBUF = (VF & VF_2BUF) ? 2 : 3;     // LDS tile buffers; tiles 0..NBUF-2 ahead are in flight
  constexpr int PD = NBUF - 1;                     // prefetch distance in tiles
  constexpr int PIECES = TILE_BYTES / 1024;        // 1 KiB DMA pieces per tensor per tile
  constexpr int PPW = PIECES / NW;                 // pieces per wave per tensor
  constexpr int DS = D / 16;
  constexpr int DT = D / 32;
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  constexpr bool WIDE = D > 128;
  static_assert(PPW >= 1 && PPW * NW == PIECES, "tile does not split into whole DMA pieces per wave");
  static_assert(!WIDE || (NW == 4 && AB == 0), "the 256-wide form: four waves, no ablations");

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
  // split-KV in ONE launch (tfa_fwd_splitkv): the grid carries nsplit copies of the work items; copy sp handles the key
  // chunk [sp*chunk, sp*chunk + nk) with the causal shift reduced by the chunk offset and writes its fp32 partial O and
  // LSE at sp * part stride.  nsplit <= 1: one chunk = the whole K/V tensor.
  const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
  const int nitems0 = p.nbh * p.nwork;
  const int nitems = nitems0 * nsplit;

  // ---- per-lane DMA source offsets (tile 0); the LDS destination of piece pc is pc*1024 + lane*16
  int k_src[PPW], v_src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave * PPW + i;
    {  // K: row-major, chunk position c' holds source chunk c' ^ swz(row)
      const int row = pc * (1024 / (D * 2)) + lane / CPR;
      const int cpos = lane % CPR;
      const int kch = cpos ^ k_swz<D>(row);            // source chunk of this lane; chunks beyond the valid head dim read as zeros
      k_src[i] = kch * 8 < p.dv ? row * (int)p.ks_n * 2 + (kch << 4) : (int)TFA_OOB;
    }
    {  // V: invert v_lds_off(): LDS offset -> (key, 16-byte chunk)
      const int o = pc * 1024 + lane * 16;
      const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
      const int dt = sub % DT, sh = sub / DT;
      const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
      v_src[i] = (dt * 4 + pcs) * 8 < p.dv ? key * (int)p.vs_n * 2 + ((dt * 4 + pcs) << 4) : (int)TFA_OOB;
    }
  }
  const int k_tile_stride = BN * (int)p.ks_n * 2;
  const int v_tile_stride = BN * (int)p.vs_n * 2;

  const int k_rd_base = qi * (D * 2);
  const int k_rd_swz = k_swz<D>(qi);
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const float sc = p.scale_log2;

  // ---- the block stream ----------------------------------------------------------------------
  struct Blk {
    int bh, wi, mb, nt;
    int sp, nk, shift;          // key chunk index, keys in the chunk, causal shift against the chunk's local key index
    __amdgpu_buffer_rsrc_t q_rs, k_rs, v_rs;
  };
  auto decode = [&](int item_all, int pass, Blk& k) {
    k.sp = item_all / nitems0;
    const int item = item_all - k.sp * nitems0;
    k.nk = p.Nk;
    k.shift = p.shift;
    if (nsplit > 1) {
      const int rest = p.Nk - k.sp * p.chunk;
      k.nk = rest < p.chunk ? (rest > 0 ? rest : 0) : p.chunk;
      k.shift = p.shift - k.sp * p.chunk;
    }
    if ((p.nbh & 7) == 0) {          // heads of one XCD stay together (item & 7 == blockIdx & 7 when G % 8 == 0)
      const int x = item & 7, s = item >> 3;
      k.bh = x + 8 * (s / p.nwork);
      k.wi = s % p.nwork;
    } else {
      k.bh = item / p.nwork;
      k.wi = item % p.nwork;
    }
    if (PAIR) k.mb = pass == 0 ? (p.nmb - 1 - k.wi) : k.wi;     // heavy block first, then the light one
    else k.mb = CAUSAL ? (p.nmb - 1 - k.wi) : k.wi;
    int kv_end = k.nk;
    if (CAUSAL) {
      const int lim = k.mb * BM + BM + k.shift;                  // one past the last key any row of the block sees
      kv_end = lim < kv_end ? lim : kv_end;
    }
    k.nt = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
    const int b = k.bh / p.H, h = k.bh - b * p.H, hk = h / (p.H / p.Hk);
    k.q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.q) + b * p.qs_b + h * p.qs_h), 0, (unsigned)p.q_bytes, 0x00020000);
    unsigned kb = (unsigned)p.k_bytes, vb = (unsigned)p.v_bytes;
    long long koff = 0, voff = 0;
    if (nsplit > 1) {                                            // descriptor over the chunk only: OOB rows read as zeros
      koff = (long long)k.sp * p.chunk * p.ks_n;
      voff = (long long)k.sp * p.chunk * p.vs_n;
      // (extent of nk rows of the VALID width p.dv: with the kernel's width D here, a head dim below D would leave the first
      //  row behind the chunk readable — for the last chunk of the last head that is memory behind the tensor, and P = 0
      //  times whatever lies there is NaN as soon as it is not finite; found by tools/fuzz_fwd.py --decode)
      kb = k.nk > 0 ? (unsigned)(((long long)(k.nk - 1) * p.ks_n + p.dv) * 2) : 0u;
      vb = k.nk > 0 ? (unsigned)(((long long)(k.nk - 1) * p.vs_n + p.dv) * 2) : 0u;
    }
    k.k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.k) + b * p.ks_b + hk * p.ks_h + koff), 0, kb, 0x00020000);
    k.v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const T*>(p.v) + b * p.vs_b + hk * p.vs_h + voff), 0, vb, 0x00020000);
  };
  // VF_DMA_NT: every K/V byte is read by exactly one workgroup (one query block per head, GQA rows packed) AND the cache is larger
  // than the 256 MB memory-side cache can keep from one decode step to the next (the host decides: tfa_api.hip) -> non-temporal
  // loads: +8..16 % on caches of 1 GB; caches that fit are served faster without the hint (-15 %: profiles/r03_decode_nt_ab.txt)
  auto dma_issue = [&](const Blk& k, int j, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave * PPW + i;
      if constexpr ((VF & VF_DMA_NT) != 0) {
        lds_dma16_nt(k.k_rs, lds_base + buf * TILE_BYTES + pc * 1024, k_src[i] + j * k_tile_stride);
        lds_dma16_nt(k.v_rs, lds_base + (NBUF + buf) * TILE_BYTES + pc * 1024, v_src[i] + j * v_tile_stride);
      } else {
        lds_dma16(k.k_rs, lds_base + buf * TILE_BYTES + pc * 1024, k_src[i] + j * k_tile_stride);
        lds_dma16(k.v_rs, lds_base + (NBUF + buf) * TILE_BYTES + pc * 1024, v_src[i] + j * v_tile_stride);
      }
    }
  };
  X8 qf[DS];
  // request a block's first two K/V tiles and its Q fragments (nothing is waited for here)
  auto prefetch = [&](const Blk& k) {
    if (k.nt > 0) dma_issue(k, 0, 0);
    if (PD > 1 && k.nt > 1) dma_issue(k, 1, 1);
    const int row = k.mb * BM + wave * 32 + qi;
    const int qoff = row * (int)p.qs_n * 2 + hi * 16;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(k.q_rs, (2 * s + hi) * 8 < p.dv ? qoff + s * 32 : (int)TFA_OOB, 0, 0);
      qf[s] = __builtin_bit_cast(X8, t);
    }
  };

  int item = blockIdx.x, pass = 0, nt_total = 0;
  bool first = true;
  if (item >= nitems) return;
  Blk cur;
  decode(item, pass, cur);
  prefetch(cur);

  while (true) {
    const int nt = cur.nt;
    nt_total += nt;
    const int wave_row0 = cur.mb * BM + wave * 32;
    const int my_row = wave_row0 + qi;

    f32x16 oacc[WIDE ? 1 : DT];
    if constexpr (WIDE) g_zero();
    else {
#pragma unroll
      for (int d = 0; d < (WIDE ? 1 : DT); ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    }
    float m_run = -1e30f;
    float l_run = 0.f;

    // tiles 0/1 and Q have been requested (prologue, or beside the previous block's epilogue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      if (WIDE) asm volatile("" : "+a"(qf[s])); else asm volatile("" : "+v"(qf[s]));
    }
    asm volatile("s_barrier" ::: "memory");
    if (p.trace && first) t_pro = __builtin_amdgcn_s_memtime();

    const int shift = cur.shift;
    const int wave_last_tile = CAUSAL ? ((wave_row0 + 31 + shift) >= 0 ? (wave_row0 + 31 + shift) / BN : -1) : (nt - 1);

    auto tile_body = [&](int j, int buf) {
      // tile j+2 goes into the buffer tile j-1 just vacated
      const bool more = (j + PD < nt) && !(AB & AB_NOSTAGE);
      if (more) dma_issue(cur, j + PD, (buf + PD) % NBUF);

      if (j <= wave_last_tile) {
        const char* kb = kl + buf * TILE_BYTES;
        const char* vb = vl + buf * TILE_BYTES;

        f32x16 sacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
        if constexpr (WIDE) {
          // k-steps in groups of four, the fragments of a group read one group ahead of the asm MFMAs that use them
          constexpr int GS = 4, NG = DS / GS;
          X8 kq[2][GS][2];
          auto rdk = [&](int gq, int buf) {
#pragma unroll
            for (int i = 0; i < GS; ++i)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                kq[buf][i][t] = __builtin_bit_cast(X8, lds_read_b128(kb, k_rd_base + t * 32 * (D * 2) + (((2 * (gq * GS + i) + hi) ^ k_rd_swz) << 4)));
          };
          rdk(0, 0);
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) rdk(gq + 1, (gq + 1) & 1);
#pragma unroll
            for (int i = 0; i < GS; ++i)
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                if (gq == 0 && i == 0) E::template mfma_bacc<true>(kq[0][0][t], qf[0], sacc[t]);      // (sacc was just zeroed by VALU moves)
                else E::template mfma_bacc<false>(kq[gq & 1][i][t], qf[gq * GS + i], sacc[t]);
              }
          }
          mfma_drain(sacc[0]);
          asm volatile("" : "+v"(sacc[1]));
        } else {
          X8 kf[DS][2];
#pragma unroll
          for (int s = 0; s < DS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int off = k_rd_base + t * 32 * (D * 2) + (((2 * s + hi) ^ k_rd_swz) << 4);
              if (AB & AB_NOKREAD) kf[s][t] = qf[(s + t) % DS];
              else kf[s][t] = __builtin_bit_cast(X8, lds_read_b128(kb, off));
            }
          if (VF & VF_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int s = 0; s < DS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              if (AB & AB_NOQK) asm volatile("" ::"v"(kf[s][t]));
              else sacc[t] = E::mfma(kf[s][t], qf[s], sacc[t]);
            }
          if (VF & VF_PRIO) __builtin_amdgcn_s_setprio(0);
        }

        auto rdv = [&](int s, int d) -> s16x8 {
          const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
          const s16x4 lo = lds_read_tr16_b64(a), hh = lds_read_tr16_b64(a + 256);
          return __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        s16x8 vq[WIDE ? 2 : 1][WIDE ? DT : 1];                // WIDE: the DT fragments of one 16-key slot, read one slot ahead
        if constexpr (WIDE) {
#pragma unroll
          for (int d = 0; d < DT; ++d) vq[0][d] = rdv(0, d);
        }
        s16x8 vfr[WIDE ? 1 : DT][WIDE ? 1 : 4];
#pragma unroll
        for (int s = 0; s < (WIDE ? 0 : 4); ++s)
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const char* a = vb + v_rd_base + (s * 2 * DT << 9) + (d << 9);
            if (AB & AB_NOVREAD) {
              vfr[d][s] = __builtin_bit_cast(s16x8, qf[(d + s) % DS]);
            } else {
              s16x4 lo = lds_read_tr16_b64(a);
              s16x4 hh = lds_read_tr16_b64(a + 256);
              vfr[d][s] = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
            }
          }

        const int key0 = j * BN;
        bool need_mask = (key0 + BN > cur.nk);
        if (CAUSAL) need_mask = need_mask || (key0 + BN - 1 > wave_row0 + shift);
        if (need_mask) {
          int lim = cur.nk - 1;
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

        float mloc = sacc[0][0];
        if (!(AB & AB_NOSM)) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[t][r]);
          mloc = pair_max(mloc);
        }
        const float m_new = (AB & AB_NOSM) ? m_run : fmaxf(m_run, mloc);
        const bool changed = (m_new != m_run);
        if (__any(changed)) {
          const float alpha = fast_exp2((m_run - m_new) * sc);
          l_run *= alpha;
          if constexpr (WIDE) g_scale(alpha);
          else {
#pragma unroll
            for (int d = 0; d < (WIDE ? 1 : DT); ++d)
#pragma unroll
              for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
          }
        }
        m_run = m_new;
        const float msc = m_new * sc;
        X8 pk[4];
        float lsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float e;
            if (AB & AB_NOSM) e = sacc[t][r];
            else e = fast_exp2(fmaf(sacc[t][r], sc, -msc));
            if (!(AB & AB_NOSM)) lsum[r & 3] += e;
            pk[t * 2 + (r >> 3)][r & 7] = (T)e;
          }
        l_run += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);

        if (VF & VF_PRIO) __builtin_amdgcn_s_setprio(1);
        if constexpr (WIDE) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) {
#pragma unroll
              for (int d = 0; d < DT; ++d) vq[(s + 1) & 1][d] = rdv(s + 1, d);
            }
#pragma unroll
            for (int d = 0; d < DT; ++d) {
              if (d == 0) g_mfma_d<T, true>(d, __builtin_bit_cast(X8, vq[s & 1][d]), pk[s]);
              else g_mfma_d<T, false>(d, __builtin_bit_cast(X8, vq[s & 1][d]), pk[s]);
            }
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int d = 0; d < (WIDE ? 1 : DT); ++d) {
              if (AB & AB_NOPV) asm volatile("" ::"v"(vfr[d][s]), "v"(pk[s]));
              else oacc[d] = E::mfma(__builtin_bit_cast(X8, vfr[d][s]), pk[s], oacc[d]);
            }
        }
        if (VF & VF_PRIO) __builtin_amdgcn_s_setprio(0);
      }

      // tile j+1 must have landed (this wave's pieces; the barrier covers everyone else's), and
      // every wave must be done reading tile j before tile j+3 overwrites it.  Counted wait: the
      // 2*PPW pieces of tile j+2 issued above may stay in flight.
      if (more && PD > 1) {
        if (PPW == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (PPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (AB & 256) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // AB_NOBARRIER (timing only)
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    if (NBUF == 3) {
      for (int j = 0; j < nt; j += 3) {
        tile_body(j, 0);
        if (j + 1 < nt) tile_body(j + 1, 1);
        if (j + 2 < nt) tile_body(j + 2, 2);
      }
    } else {
      for (int j = 0; j < nt; j += 2) {
        tile_body(j, 0);
        if (j + 1 < nt) tile_body(j + 1, 1);
      }
    }
    if (p.trace && first) t_loop = __builtin_amdgcn_s_memtime();
    first = false;

    // ---- next block of the stream ----------------------------------------------------------------
    const int cur_bh = cur.bh;
    const long long o_part = (long long)cur.sp * p.o_part_stride, lse_part = (long long)cur.sp * p.lse_part_stride;   // 0 unless split
    constexpr bool LDS_EPI = !F32OUT && (VF & VF_LDSEPI);   // 16-bit O goes out through LDS as whole rows
    bool have_next;
    if (PAIR && pass == 0 && (p.nmb - 1 - cur.wi) != cur.wi) {
      pass = 1;
      have_next = true;
    } else {
      pass = 0;
      item += gridDim.x;
      have_next = item < nitems;
    }
    if (have_next && !LDS_EPI) {
      decode(item, pass, cur);
      prefetch(cur);            // LDS buffers are free: every wave passed the last tile's barrier
    }

    // ---- epilogue of the block just finished ------------------------------------------------------
    const int ob = cur_bh / p.H, oh = cur_bh - ob * p.H;
    float og[WIDE ? DT : 1][16];                           // WIDE: O read out of the hand-owned AccVGPRs
    if constexpr (WIDE) {
#pragma unroll
      for (int d = 0; d < DT; ++d) g_read_d(d, og[d]);
    }
    auto ov = [&](int d, int i) -> float { return WIDE ? og[WIDE ? d : 0][i] : oacc[WIDE ? 0 : d][i]; };
    const float l_tot = pair_sum(l_run);
    const bool empty = !(l_tot > 0.f);
    const float inv = empty ? 1.f : 1.f / l_tot;
    if (p.lse != nullptr && hi == 0 && my_row < p.Nq) {
      const float lse = empty ? INFINITY : (m_run * p.scale + __builtin_amdgcn_logf(l_tot) * 0.6931471805599453f);
      p.lse[lse_part + (long long)cur_bh * p.Nq + my_row] = lse;
    }
    if (F32OUT) {
      float* obase = reinterpret_cast<float*>(p.o) + o_part + ob * p.os_b + oh * p.os_h;
      auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
      const int ooff = my_row * (int)p.os_n * 4 + hi * 16;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v4 = {ov(d, 4 * g + 0) * inv, ov(d, 4 * g + 1) * inv, ov(d, 4 * g + 2) * inv, ov(d, 4 * g + 3) * inv};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), o_rs, d * 32 + g * 8 + hi * 4 < p.dv ? ooff + (d * 32 + g * 8) * 4 : (int)TFA_OOB, 0, 0);
        }
    } else if (LDS_EPI) {
      // Each lane holds 4-element pieces of ONE row scattered over 16 registers groups: stored directly that is
      // 16 eight-byte stores per lane to 32 different rows per instruction.  Instead the wave transposes its
      // 32 x D tile through its own slice of the (now idle) K buffers — 16-byte chunk index XOR row, as for K —
      // and writes whole rows: 1 KiB contiguous per store instruction.
      T* obase = reinterpret_cast<T*>(p.o) + o_part + ob * p.os_b + oh * p.os_h;
      auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
      typedef __attribute__((ext_vector_type(4))) T t4;
      char* const ow = smem + wave * (32 * D * 2);
      constexpr int CH = D / 8;                      // 16-byte chunks per row
      const int osw = (CH == 16) ? (qi & 15) : (qi & 7);
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          t4 v4 = {(T)(ov(d, 4 * g + 0) * inv), (T)(ov(d, 4 * g + 1) * inv), (T)(ov(d, 4 * g + 2) * inv), (T)(ov(d, 4 * g + 3) * inv)};
          const int c = d * 4 + g;
          *reinterpret_cast<u32x2*>(ow + qi * (D * 2) + ((c ^ osw) << 4) + hi * 8) = __builtin_bit_cast(u32x2, v4);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no barrier needed
      constexpr int RPI = 64 / CH;                   // rows per store instruction (4 at D=128, 8 at D=64)
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + lane / CH, cpos = lane % CH;
        const int c = cpos ^ ((CH == 16) ? (r & 15) : (r & 7));
        u32x4 v = *reinterpret_cast<const u32x4*>(ow + r * (D * 2) + (cpos << 4));
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, c * 8 < p.dv ? (wave_row0 + r) * (int)p.os_n * 2 + (c << 4) : (int)TFA_OOB, 0, 0);
      }
      if (have_next) {
        // the next block's DMA will overwrite these slices: every wave must have read its rows back
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        decode(item, pass, cur);
        prefetch(cur);
      }
    } else {
      T* obase = reinterpret_cast<T*>(p.o) + o_part + ob * p.os_b + oh * p.os_h;
      auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, (unsigned)p.o_bytes, 0x00020000);
      const int ooff = my_row * (int)p.os_n * 2 + hi * 8;
      typedef __attribute__((ext_vector_type(4))) T t4;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          t4 v4 = {(T)(ov(d, 4 * g + 0) * inv), (T)(ov(d, 4 * g + 1) * inv), (T)(ov(d, 4 * g + 2) * inv), (T)(ov(d, 4 * g + 3) * inv)};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v4), o_rs, d * 32 + g * 8 + hi * 4 < p.dv ? ooff + (d * 32 + g * 8) * 2 : (int)TFA_OOB, 0, 0);
        }
    }
    if (!have_next) break;
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
      t[7] = (unsigned long long)blockIdx.x;
    }
  }
}

}  // namespace tfa
