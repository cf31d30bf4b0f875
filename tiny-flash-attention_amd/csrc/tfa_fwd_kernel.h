// tfa_fwd_kernel.h — what every forward / backward kernel of this library shares: the kernel arguments (KArgs), the MFMA element
// traits, the LDS layouts of the K and V tile images, the half-wave exchange and the windowed buffer descriptors (gfx950, CDNA4).
// (The first bring-up kernel that used to live here is experiments/csrc/tfa_fwd_kernel_bringup.h.)
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

// Division of a non-negative int below 2^31 by a launch constant: (umulhi(n, m) + n) >> l — three scalar instructions where hipcc's signed
// division is about thirty (round 4: a workgroup's start-up is bound by instruction issue, docs/LABLOG.md L-10)
struct FastDiv {
  unsigned m, l;
};
static inline FastDiv fastdiv_of(unsigned d) {     // d >= 1
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  FastDiv f;
  f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.l = l;
  return f;
}
static __device__ __forceinline__ int fd_div(int n, FastDiv f) { return (int)((__umulhi((unsigned)n, f.m) + (unsigned)n) >> f.l); }

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
  // work-item decode of fwd_kernel_il, filled by launch_common (tfa_launch.h: fill_decode) from the fields above
  // One branch-free form for the three orders (so that hipcc fetches every kernel argument the decode needs in one burst):
  //   x = rr ? id & 7 : 0, s = rr ? id >> 3 : id;  sq = s / wa, r = s % wa;  kg = rr ? x + 8 sq : sq;  rq = r / nwork, wi = r % nwork;
  //   b = kg / wd, m = kg % wd;  h = m * wg + rq;  hk = h / G
  //   GQA with (B * Hk) % 8 == 0: K/V heads round-robin over the XCDs, the G query heads of one stay on its XCD: rr = 1, wa = G * nwork, wd = Hk, wg = G
  //   B * H % 8 == 0: heads round-robin over the XCDs: rr = 1, wa = nwork, wd = H, wg = 1  (rq = 0);   else plain (b,h)-major: rr = 0, same
  int rr, wa, wd, wg, G;
  FastDiv fd_wa, fd_wd, fd_nwork, fd_g;
  int kv_stream;    // K and V together reach 768 MiB — a cache the 256 MB memory-side cache cannot keep until the next call: decode
                    // kernels whose K/V tiles no other workgroup reads may stream them with the non-temporal hint (set by the host)
};

template <typename T> struct Elem;
template <> struct Elem<__bf16> {
  using x8 = bf16x8;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // c += a . b with the B operand read from AccVGPRs and the accumulator in architectural VGPRs — for kernels whose register
  // budget is "accumulators and resident operands in AccVGPRs, everything VALU touches in VGPRs" (one wave per SIMD, 512
  // registers).  hipcc picks ONE form for every MFMA of a kernel; with the builtin the VALU-consumed accumulators land in AccVGPRs
  // too and everything is copied around.  Inline asm: nothing pads the MFMA -> VALU hazard behind it (mfma_drain below).
  // PAD: the two wait states a VALU-written operand (the freshly zeroed c) needs in front of the MFMA.
  template <bool PAD> static __device__ __forceinline__ void mfma_bacc(x8 a, x8 b, f32x16& c) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
  }

};
// wait states between the last asm MFMA that wrote c and the first instruction that reads it (8-pass MFMA: 12)
static __device__ __forceinline__ void mfma_drain(f32x16& c) { asm volatile("s_nop 12" : "+v"(c)); }
template <> struct Elem<_Float16> {
  using x8 = f16x8;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  // (see Elem<__bf16>::mfma_bacc)
  template <bool PAD> static __device__ __forceinline__ void mfma_bacc(x8 a, x8 b, f32x16& c) {
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
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

}  // namespace tfa
