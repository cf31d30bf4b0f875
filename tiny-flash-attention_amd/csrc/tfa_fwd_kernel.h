// tfa_fwd_kernel.h — the fused FlashAttention-2 forward tile loop for gfx950 (MI355X, CDNA4).
//
// Hand-written for 64-wide wavefronts and the 32x32x16 bf16/f16 MFMA; not derived from the
// reference's CuTe/SM80 kernel.  What it computes is the reference's hot loop
// (flash_attention_cutlass/csrc/flash_attention.cu:373-685): per query block, stream K/V
// blocks, S = QK^T, online softmax, O += PV, normalise, write O and LSE.
//
// Design (see DESIGN.md for the numbers):
//   * workgroup = NW waves; wave w owns 32 query rows; KV block = 64 keys.
//   * "swapped" first GEMM: S^T = K Q^T, so the 32x32 MFMA result puts ONE query row in each
//     lane (column = lane&31) and keys in registers -> row max / row sum are in-lane
//     reductions plus ONE half-wave exchange (v_permlane32_swap); running max/sum live in VGPRs.
//   * second GEMM O^T += V^T P^T with P^T taken straight from the S^T accumulator registers
//     (rounded to 16 bit): the MFMA k-index is permuted consistently for P and V, so no
//     cross-lane movement of P is needed at all.
//   * K tile in LDS row-major with a 16-byte-chunk XOR swizzle (conflict-free ds_read_b128);
//     V tile in LDS as [8-key][32-col] sub-tiles read with ds_read_b64_tr_b16 (hardware
//     transpose) — each half-wave reads one contiguous 256 B span.
//   * K/V global->register->LDS staging is split (issue loads for tile j+1 before the compute
//     of tile j, write them to the other LDS buffer after it): one barrier per KV tile.
//   * buffer loads/stores with hardware bounds checking give ragged N for free
//     (out-of-range rows read as 0 / are not written).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Kernel arguments (device view of tfa_fwd_params; strides in ELEMENTS).
struct KArgs {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  int B, H, Hk, Nq, Nk;
  int shift;        // causal: key j of THIS K/V tensor is visible to row i iff j <= i + shift
                    // (= Nk - Nq bottom-right aligned; less the chunk offset when K/V is a chunk of a longer sequence)
  int nsplit;       // split-KV in one launch (tfa_fwd_splitkv; LDS-DMA kernel only): number of key chunks, 0/1 = none
  int chunk;        // keys per chunk (multiple of 64)
  long long o_part_stride, lse_part_stride;   // elements between the partial results of consecutive chunks
  int nmb;          // number of query blocks per (b,h)
  int nwork;        // work items per (b,h): nmb, or ceil(nmb/2) when causal blocks are paired
  int nbh;          // B*H
  long long qs_b, qs_h, qs_n;
  long long ks_b, ks_h, ks_n;
  long long vs_b, vs_h, vs_n;
  long long os_b, os_h, os_n;
  unsigned long long q_bytes, k_bytes, v_bytes, o_bytes;   // extent of one (b,h) slice in bytes.  Kernels use ONE descriptor per slice
                    // (< 2 GiB); the il kernels also exist in a WINDOWED instantiation (VF_IL_WINDOWED: rsrc_at, one query
                    // block / one K/V tile per descriptor) that tfa_api.hip launches when a slice is larger
  int big;          // some slice does not fit one descriptor: launch the windowed instantiation
  int row_mod;      // > 0: GQA query heads packed as rows (tfa_api.hip: pack_gqa_rows) in a CAUSAL problem — row r of the block is
                    // query position r % row_mod of one of the packed heads; 0: row r is position r
  float scale;      // softmax_scale
  float scale_log2; // softmax_scale * log2(e)
  int grid;         // workgroups launched (persistent kernels walk work items with this stride)
  unsigned long long* trace;  // debug: 8 x u64 per workgroup (cycle stamps), or nullptr
  int dv;           // valid head dim (<= the kernel's compile-time D, a multiple of 8): the 16-byte chunks of a row beyond dv are
                    // read as zeros (their LDS-DMA lanes / Q loads are pointed outside the buffer) and never stored
  int dbg;          // bring-up flags (tfa_debug_set_flags; 0 in normal use).  128: the trace stamps describe the workgroup's
                    // SECOND pass (t[0] = its start) instead of the first; low bits: tfa_fwd_kernel_x4.h
};

template <typename T> struct Elem;
template <> struct Elem<__bf16> {
  using x8 = bf16x8;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<_Float16> {
  using x8 = f16x8;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

typedef __attribute__((address_space(3))) char lds_char;

static __device__ __forceinline__ u32x4 lds_read_b128(const char* base, int off) {
  return *reinterpret_cast<const u32x4*>(base + off);
}
static __device__ __forceinline__ void lds_write_b128(char* base, int off, u32x4 v) {
  *reinterpret_cast<u32x4*>(base + off) = v;
}
// ds_read_b64_tr_b16: within each 16-lane group the 16 lanes x 4 halfwords that the lanes
// address are transposed: lane i receives halfword (i&3) of the four lanes 4j+(i>>2), j=0..3.
static __device__ __forceinline__ s16x4 lds_read_tr16_b64(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
}

static __device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Exchange between the two half-waves (lane l <-> lane l^32) with ONE v_permlane32_swap.
// After `v_permlane32_swap a, b` (a: upper half <-> b: lower half) with a == b == x on entry:
//   a = {x_lo, x_lo},  b = {x_hi, x_hi}   in lanes {0-31, 32-63}.
// Written as inline asm: with identical operands hipcc (ROCm 7.2) folds the builtin's two results
// into one register and the reduction silently degenerates to max(x,x) / x+x.  The s_nop covers the
// "VALU write -> v_permlane read" hazard (2 wait states), which nothing pads inside an asm statement.
static __device__ __forceinline__ void half_swap(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}
static __device__ __forceinline__ float pair_max(float x) {
  float lo, hi;
  half_swap(x, lo, hi);
  return fmaxf(lo, hi);
}
static __device__ __forceinline__ float pair_sum(float x) {
  float lo, hi;
  half_swap(x, lo, hi);
  return lo + hi;
}

// ---- LDS layouts -------------------------------------------------------------------------
// K tile [64 keys][D] row-major, 16-byte chunk index XOR-swizzled by the row so that the 16
// lanes of a ds_read_b128 service group (16 different keys, same chunk) hit 16 different
// 16-byte slots of the 256-byte bank row.
template <int D> static __device__ __forceinline__ int k_swz(int row) {
  return D >= 128 ? (row & 15) : ((row >> 1) & 7);   // (D = 256: 32 chunks per row, the XOR stays inside a 16-chunk half)
}
template <int D> static __device__ __forceinline__ int k_lds_off(int row, int chunk) {
  return row * (D * 2) + ((chunk ^ k_swz<D>(row)) << 4);
}
// V tile [64 keys][D]: sub-tiles of [8 key-rows][32 cols] (512 B).  Key k of a 16-key slot s
// goes to sub-tile half (k>>2)&1 (the half-wave that consumes it) and row ((k>>3)&1)*4+(k&3).
// With this order the MFMA k-index of the PV product is {0-3,8-11 | 4-7,12-15} per half-wave,
// which is exactly how the S^T accumulator hands each lane its P values.
template <int D> static __device__ __forceinline__ int v_lds_off(int key, int chunk /*16B chunk in row*/) {
  const int s = key >> 4, kk = key & 15;
  const int hi = (kk >> 2) & 1, half = kk >> 3, r4 = kk & 3;
  return (((s * 2 + hi) * (D / 32) + (chunk >> 2)) << 9) + ((half * 4 + r4) << 6) + ((chunk & 3) << 4);
}

// byte offset that every buffer descriptor of these kernels rejects (slices are < 2 GiB): reads return 0, stores are dropped.
// Used with UNSIGNED arithmetic: OOB + any tile offset (< 2^31) stays >= 2^31 without wrapping.
constexpr unsigned TFA_OOB = 0x80000000u;

// Buffer descriptor over [base + off, base + total): the window a kernel addresses with 32-bit offsets.  At most 2 GiB - 1 bytes
// are visible through it; what lies beyond `total` reads as zeros / drops stores exactly as with a whole-slice descriptor.
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void* base, unsigned long long total, unsigned long long off) {
  // 32-bit scalar pieces only (a 64-bit compare goes through the VALU): rem = total - off as a signed 64-bit value
  const unsigned long long rem = total - off;
  const unsigned hi = (unsigned)(rem >> 32), lo = (unsigned)rem;
  unsigned n = lo < 0x7fffffffu ? lo : 0x7fffffffu;
  n = hi ? 0x7fffffffu : n;
  n = ((int)hi < 0) ? 0u : n;                        // off beyond the slice: nothing visible
  return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(base) + off), 0, n, 0x00020000);
}

// Variant flags
constexpr int VF_TRREAD = 1;     // V fragments by ds_read_b64_tr_b16 (else 16-bit gathers)
constexpr int VF_NOSKIP = 2;     // always rescale O (no exact alpha==1 skip)
constexpr int VF_PAIR = 4;       // causal: one workgroup walks query blocks (nmb-1-i) and (i): equal work per workgroup
constexpr int VF_KPRE = 8;       // read all K fragments of a tile into registers before the QK^T MFMAs
constexpr int VF_W64 = 8192;     // 64 rows per wave, AGPR-pinned O (tfa_fwd_kernel_w64.h)
constexpr int VF_SWP = 128;      // software-pipelined DMA kernel (tfa_fwd_kernel_swp.h)
constexpr int VF_DMA = 64;       // LDS-DMA staging kernel (tfa_fwd_kernel_dma.h)
constexpr int VF_PP = 32;        // ping-pong schedule (tfa_fwd_kernel_pp.h)
constexpr int VF_VPRE_SHIFT = 8;      // ping-pong kernel, bits 8..10: number of 32-wide d-tiles whose V fragments are read in the first half
constexpr int VF_NOPQK_SHIFT = 16;    // ping-pong kernel, bits 16..20: 0 = off, n = "s_nop n-1" after every QK^T MFMA
constexpr int VF_NOPPV_SHIFT = 21;    // ping-pong kernel, bits 21..25: same after every PV MFMA
constexpr int VF_VPRE = 16;      // issue all V fragment reads before the softmax, PV runs from registers

// Ablation bits (timing experiments only — results are WRONG with any bit set; never dispatched by tfa_fwd)
constexpr int AB_NOQK = 1, AB_NOPV = 2, AB_NOEXP = 4, AB_NOSM = 8, AB_NOKREAD = 16, AB_NOVREAD = 32, AB_NOSTAGE = 64, AB_NOCVT = 128;

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
