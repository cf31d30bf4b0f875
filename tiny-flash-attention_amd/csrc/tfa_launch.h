// tfa_launch.h — host-side dispatch table shared by the per-(dtype, head-dim) instantiation units.
#pragma once
// The product library carries six kernels (kVariants); every measured dead end and A/B arm lives in kExperimentalVariants and is compiled only
// with -DTFA_EXPERIMENTAL (make EXPERIMENTAL=1).  Without the flag those variant numbers are rejected (TFA_ERR_VARIANT).
#include <hip/hip_runtime.h>
#include "tfa_host_util.h"
#include "tfa_fwd_kernel.h"
#include "tfa_fwd_kernel_dma.h"
#include "tfa_fwd_kernel_il.h"
#include "tfa_fwd_kernel_x4.h"
#if defined(TFA_EXPERIMENTAL)   // measured dead ends, kept out of the package: experiments/csrc (the Makefile adds the include path)
#include "tfa_fwd_kernel_bringup.h"
#include "tfa_fwd_kernel_pp.h"
#include "tfa_fwd_kernel_swp.h"
#include "tfa_fwd_kernel_w64.h"
#endif

namespace tfa {

struct Variant {
  int id;           // the number tfa_set_variant / tfa_fwd_variant speak (stable across rounds: tools, tests and DESIGN.md cite them)
  const char* name;
  int nw;   // waves per workgroup
  int vf;   // VF_* flags
  int rb;   // 32-row query blocks per wave
};

// ---- the product library: seven kernels -------------------------------------------------------------------------------------
constexpr int kDefaultVariant = 30;       // il8: 256-row blocks, issue-interleaved, 16-bit O through a separate LDS region
constexpr int kSmallGridVariant = 32;     // il4: 128-row query blocks, two workgroups per CU
constexpr int kX4D256Variant = 34;        // x4-d256: the only kernel for head dims above 128
constexpr int kKSplitVariant = 36;        // il8-ksplit: grids of at most one 128-row block per CU
constexpr int kKSplitPairVariant = 37;    // il8-ksplit-pair: causal grids of at most two 128-row blocks per CU
constexpr int kExactVariant = 38;         // exact-il8: the il8 kernel with the exact running row maximum — TFA_FWD_EXACT_MAX wherever the default would run il8 (round 5)
constexpr int kSplitVariant = 17;         // dma4-2buf: the kernel whose grid can carry key chunks (tfa_fwd_splitkv), exact running max (TFA_FWD_EXACT_MAX)
static const Variant kVariants[] = {
    {17, "dma4-pair-2buf (LDS-DMA, burst-structured, exact running max; 128-row blocks, two LDS buffers: two workgroups per CU)", 4, VF_DMA | VF_PAIR | VF_2BUF, 1},
    {30, "il8-pair-dmaspread-epi (issue-interleaved, 8 waves; O leaves through a separate LDS region as whole rows, 16-byte stores; causal pairs: the light "
         "pass is requested before the heavy pass's O stores; a workgroup's first Q rows arrive through LDS)", 8,
     VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD | VF_IL_EPI | VF_IL_PREF2 | VF_IL_QLDS, 1},
    {32, "il4-pair-epi (4 waves x2 workgroups per CU; O leaves through the idle tile buffers as whole rows)", 4, VF_DMA | VF_IL | VF_PAIR | VF_IL_EPI | VF_IL_EPI_INPLACE, 1},
    {34, "x4-d256-pair (the x4 kernel with ONE 32-row block per wave: head dims 136..256, 128-row workgroups, O leaves through the idle tile buffers as whole rows)", 4, VF_DMA | VF_IL | VF_X4 | VF_PAIR, 1},
    {36, "il8-ksplit-epi (small grids: 8 waves on one 128-row block, waves 0-3 take the even KV tiles and waves 4-7 the odd ones, merged through LDS)", 8,
     VF_DMA | VF_IL | VF_IL_DMASPREAD | VF_IL_EPI | VF_IL_EPI_INPLACE | VF_IL_KSPLIT, 1},
    {37, "il8-ksplit-pair-epi (the key-split kernel with causal blocks paired heavy+light per workgroup: two 128-row blocks per CU)", 8,
     VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD | VF_IL_EPI | VF_IL_EPI_INPLACE | VF_IL_KSPLIT, 1},
    {38, "exact-il8-pair-dmaspread-epi (variant 30's kernel with the EXACT running row maximum — the reference's rounding points of P, TFA_FWD_EXACT_MAX on grids "
         "that fill the chip: tiles in which some row of a wave saw a new maximum run a body that also re-bases O behind the QK^T MFMAs)", 8,
     VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD | VF_IL_EPI | VF_IL_PREF2 | VF_IL_QLDS | VF_IL_EXACT, 1},
};
constexpr int kNumProductVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// ---- A/B arms and measured dead ends: `make EXPERIMENTAL=1` only (tfa_set_variant rejects their numbers otherwise) -------------
constexpr int kX4Variant = 33;            // il-x4-pair-epi (the x4 kernel at head dims <= 128)
constexpr int kSeamVariant = 35;          // il8-pair-dmaspread-epi-seam
// (round 4's arms of the product kernel — numbers 38..62 — are a patch: experiments/r04_il8_arms.patch, docs/LABLOG.md L-9)
#if defined(TFA_EXPERIMENTAL)
static const Variant kExperimentalVariants[] = {
    {0, "w8-gatherV (bring-up: 16-bit LDS gathers for V, no transpose read)", 8, 0, 1},
    {1, "w8-trV (8 waves x 32 rows, ds_read_b64_tr_b16 for V)", 8, VF_TRREAD, 1},
    {2, "w4-trV (4 waves x 32 rows, 2 workgroups/CU)", 4, VF_TRREAD, 1},
    {3, "w8-trV-alwaysrescale (no exact alpha==1 skip)", 8, VF_TRREAD | VF_NOSKIP, 1},
    {4, "w8-trV-pair (causal blocks paired heavy+light per workgroup)", 8, VF_TRREAD | VF_PAIR, 1},
    {5, "w8-trV-pair-kpre-vpre (K and V fragments prefetched to registers)", 8, VF_TRREAD | VF_PAIR | VF_KPRE | VF_VPRE, 1},
    {6, "w4-trV-pair-kpre-vpre", 4, VF_TRREAD | VF_PAIR | VF_KPRE | VF_VPRE, 1},
    {7, "w8-trV-pair-vpre", 8, VF_TRREAD | VF_PAIR | VF_VPRE, 1},
    {8, "pp8-pair-vpre0 (ping-pong: two wave groups half a tile apart; V fragments read in the 2nd half)", 8, VF_PP | VF_PAIR | (0 << VF_VPRE_SHIFT), 1},
    {9, "pp8-pair-vpre2 (ping-pong; V fragments of 2 d-tiles read ahead in the 1st half)", 8, VF_PP | VF_PAIR | (2 << VF_VPRE_SHIFT), 1},
    {10, "pp8-pair-vpre4 (ping-pong; all V fragments read ahead in the 1st half)", 8, VF_PP | VF_PAIR | (4 << VF_VPRE_SHIFT), 1},
    {11, "dma8-pair (LDS-DMA staging, 3 tile buffers, counted vmcnt; 8 waves)", 8, VF_DMA | VF_PAIR, 1},
    {12, "dma4-pair (LDS-DMA staging, 3 tile buffers; 4 waves, 1 workgroup/CU by LDS)", 4, VF_DMA | VF_PAIR, 1},
    {13, "dma8-pair-setprio", 8, VF_DMA | VF_PAIR | VF_PRIO, 1},
    {14, "swp8-pair (LDS-DMA + software-pipelined loop: softmax(j) beside QK^T(j+1), PV(j) beside rowmax(j+1))", 8, VF_DMA | VF_SWP | VF_PAIR, 1},
    {15, "dma8-pair-persistent (256 workgroups walk the work items; next block prefetched behind the epilogue)", 8, VF_DMA | VF_PAIR | VF_PERSIST, 1},
    {16, "dma4-pair-persistent (128-row blocks, 256 persistent workgroups)", 4, VF_DMA | VF_PAIR | VF_PERSIST, 1},
    {18, "dma4-pair-2buf-persistent", 4, VF_DMA | VF_PAIR | VF_2BUF | VF_PERSIST, 1},
    {19, "dma8-pair-2buf (two LDS buffers, prefetch distance one tile)", 8, VF_DMA | VF_PAIR | VF_2BUF, 1},
    {20, "w64-pair (4 waves x 64 rows, one wave per SIMD, O accumulators pinned in AGPRs by inline-asm MFMA)", 4, VF_DMA | VF_W64 | VF_PAIR, 2},
    {21, "dma4-pair-2buf-ldsepi (O leaves through LDS as whole rows, 16-byte stores)", 4, VF_DMA | VF_PAIR | VF_2BUF | VF_LDSEPI, 1},
    {22, "dma8-pair-2buf-ldsepi", 8, VF_DMA | VF_PAIR | VF_2BUF | VF_LDSEPI, 1},
    {23, "pp8-vpre4-gapqk8 (ping-pong + 8-cycle issue gap behind every QK^T MFMA)", 8, VF_PP | VF_PAIR | (4 << VF_VPRE_SHIFT) | (8 << VF_NOPQK_SHIFT), 1},
    {24, "pp8-vpre4-gapqk8-gappv8", 8, VF_PP | VF_PAIR | (4 << VF_VPRE_SHIFT) | (8 << VF_NOPQK_SHIFT) | (8 << VF_NOPPV_SHIFT), 1},
    {25, "pp8-vpre4-gapqk16", 8, VF_PP | VF_PAIR | (4 << VF_VPRE_SHIFT) | (16 << VF_NOPQK_SHIFT), 1},
    {26, "il8-pair (issue-interleaved: every MFMA followed by its share of another tile's softmax; 8 waves)", 8, VF_DMA | VF_IL | VF_PAIR, 1},
    {27, "il4-pair (issue-interleaved, 4 waves, two workgroups per CU)", 4, VF_DMA | VF_IL | VF_PAIR, 1},
    {28, "il8-pair-dmaspread (LDS-DMA pieces issued between the first QK^T MFMAs)", 8, VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD, 1},
    {29, "il8-pair-dmastagger (waves 4-7 issue their LDS-DMA pieces behind the first PV MFMAs instead)", 8, VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD | VF_IL_DMASTAGGER, 1},
    {31, "il8-pair-dmaspread-epi-pref (+ the next pass's first tiles and Q requested before the epilogue, vmcnt(0) in the prologue)", 8, VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD | VF_IL_EPI | VF_IL_PREF, 1},
    {33, "il-x4-pair-epi (issue-interleaved, 4 waves x 64 rows: one wave per SIMD, O and Q in AGPRs, K ring of three LDS buffers)", 4, VF_DMA | VF_IL | VF_X4 | VF_PAIR | VF_X4_EPI, 2},
    {35, "il8-pair-dmaspread-epi-seam (the heavy pass's last tiles stream the light pass's K(0), K(1), V(0); Q before the epilogue)", 8, VF_DMA | VF_IL | VF_PAIR | VF_IL_DMASPREAD | VF_IL_EPI | VF_IL_SEAM, 1},
};
constexpr int kNumExperimentalVariants = sizeof(kExperimentalVariants) / sizeof(kExperimentalVariants[0]);
#endif
constexpr int kNumVariants = 39;          // variant numbers live in [0, kNumVariants); which of them a build carries: variant_info() != nullptr

// the table entry of a variant number, or nullptr when this build does not carry it
static inline const Variant* variant_info(int variant) {
  for (int i = 0; i < kNumProductVariants; ++i)
    if (kVariants[i].id == variant) return &kVariants[i];
#if defined(TFA_EXPERIMENTAL)
  for (int i = 0; i < kNumExperimentalVariants; ++i)
    if (kExperimentalVariants[i].id == variant) return &kExperimentalVariants[i];
#endif
  return nullptr;
}
static inline bool variant_built(int variant) { return variant_info(variant) != nullptr; }

struct LaunchGeom {
  int grid, block, lds;
};

// one translation unit per (dtype, width, causal): tfa_fwd_inst_<dtype>_<D>_c<0|1>.hip specialises launch_fwd_c
template <typename T, int D, bool CAUSAL>
hipError_t launch_fwd_c(const KArgs& a, bool f32out, int variant, hipStream_t stream, LaunchGeom* geom, bool dry);
#define TFA_FWD_UNITS(T, D)                                                                             \
  template <> hipError_t launch_fwd_c<T, D, false>(const KArgs&, bool, int, hipStream_t, LaunchGeom*, bool); \
  template <> hipError_t launch_fwd_c<T, D, true>(const KArgs&, bool, int, hipStream_t, LaunchGeom*, bool);
TFA_FWD_UNITS(__bf16, 64) TFA_FWD_UNITS(__bf16, 128) TFA_FWD_UNITS(_Float16, 64) TFA_FWD_UNITS(_Float16, 128)
#undef TFA_FWD_UNITS
template <typename T, int D>
static inline hipError_t launch_fwd(const KArgs& a, bool causal, bool f32out, int variant, hipStream_t stream, LaunchGeom* geom, bool dry) {
  return causal ? launch_fwd_c<T, D, true>(a, f32out, variant, stream, geom, dry) : launch_fwd_c<T, D, false>(a, f32out, variant, stream, geom, dry);
}

// common tail of every launcher: report the geometry, opt in to the dynamic LDS size on this device, launch, and return
// THIS launch's status (a sticky error left behind by unrelated earlier HIP calls is cleared first).
// the launch constants of the kernels' work-item decode (KArgs::wmode ..): every launch goes through launch_common, so no caller can forget them
static inline void fill_decode(KArgs* a) {
  a->G = a->Hk > 0 ? a->H / a->Hk : 1;
  if (a->G < 1) a->G = 1;
  const int nwork = a->nwork > 0 ? a->nwork : 1;
  const bool gqa_rr = a->G > 1 && ((a->B * a->Hk) & 7) == 0;
  a->rr = (gqa_rr || (a->nbh & 7) == 0) ? 1 : 0;
  if (a->dbg & 65536) a->rr = 0;     // measurement aid (tools/decode_order_pmc.sh): the plain order — a head's work items on consecutive workgroups, i.e. spread over all XCDs
  a->wa = gqa_rr ? a->G * nwork : nwork;
  a->wd = gqa_rr ? a->Hk : (a->H > 0 ? a->H : 1);
  a->wg = gqa_rr ? a->G : 1;
  a->fd_wa = fastdiv_of(a->wa);
  a->fd_wd = fastdiv_of(a->wd);
  a->fd_nwork = fastdiv_of(nwork);
  a->fd_g = fastdiv_of(a->G);
}

template <typename Kern>
static inline hipError_t launch_common(Kern kern, std::atomic<unsigned long long>& mask, int grid, int block, int lds,
                                       const KArgs& a_in, hipStream_t stream, LaunchGeom* geom, bool dry) {
  if (geom) { geom->grid = grid; geom->block = block; geom->lds = lds; }
  if (dry) return hipSuccess;
  KArgs a = a_in;
  fill_decode(&a);
  hipError_t e = set_dyn_lds_once(mask, reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a);
  return hipGetLastError();
}

// The LDS-DMA kernel 256 wide, fp32 partial output: tfa_fwd_splitkv's one-launch form for head dims above 128 (tfa_dma_inst_<dtype>_256.hip)
template <typename T>
hipError_t launch_splitkv_wide(const KArgs& a, bool causal, hipStream_t stream, LaunchGeom* geom, bool dry);

// The x4 kernel: one translation unit per (dtype, width, causal, output type) — tfa_x4_inst_<dtype>_<D>_c<0|1>_o<16|32>.hip —
// each specialising launch_x4_piece; ablate != 0 selects a timing-only ablation (builds with -DTFA_X4_ABLATE).
template <typename T, int D, bool CAUSAL, bool F32OUT>
hipError_t launch_x4_piece(const KArgs& a, int ablate, hipStream_t stream, LaunchGeom* geom, bool dry);
#define TFA_X4_PIECES(T, D)                                                                                   \
  template <> hipError_t launch_x4_piece<T, D, false, false>(const KArgs&, int, hipStream_t, LaunchGeom*, bool); \
  template <> hipError_t launch_x4_piece<T, D, false, true>(const KArgs&, int, hipStream_t, LaunchGeom*, bool);  \
  template <> hipError_t launch_x4_piece<T, D, true, false>(const KArgs&, int, hipStream_t, LaunchGeom*, bool);  \
  template <> hipError_t launch_x4_piece<T, D, true, true>(const KArgs&, int, hipStream_t, LaunchGeom*, bool);
TFA_X4_PIECES(__bf16, 64) TFA_X4_PIECES(__bf16, 128) TFA_X4_PIECES(__bf16, 256)
TFA_X4_PIECES(_Float16, 64) TFA_X4_PIECES(_Float16, 128) TFA_X4_PIECES(_Float16, 256)
#undef TFA_X4_PIECES
// the 256-wide x4 kernel with fewer than eight valid 32-column blocks (head dims 136..224): one unit per (dtype, causal, output type,
// block count) — tfa_x4_inst_<dtype>_256_c<0|1>_o<16|32>_v<5|6|7>.hip
template <typename T, bool CAUSAL, bool F32OUT, int DVB>
hipError_t launch_x4_wide(const KArgs& a, hipStream_t stream, LaunchGeom* geom, bool dry);

template <typename T, int D>
static inline hipError_t launch_x4_unit(const KArgs& a, bool causal, bool f32out, int ablate, hipStream_t stream, LaunchGeom* geom, bool dry) {
  if (causal) return f32out ? launch_x4_piece<T, D, true, true>(a, ablate, stream, geom, dry) : launch_x4_piece<T, D, true, false>(a, ablate, stream, geom, dry);
  return f32out ? launch_x4_piece<T, D, false, true>(a, ablate, stream, geom, dry) : launch_x4_piece<T, D, false, false>(a, ablate, stream, geom, dry);
}

// kernels that honour KArgs::dv (head dims below the compiled width: LDS-DMA lanes / Q loads / O stores of the missing
// 16-byte chunks go out of range): the LDS-DMA kernel, the il kernels and the x4 kernel
static inline bool supports_padded_d(int variant) {
  const Variant* v = variant_info(variant);
  return v && (v->vf & VF_DMA) && !(v->vf & (VF_SWP | VF_W64));
}

// kernels that address a (b,h) slice through windowed descriptors (rsrc_at): the slice may exceed 2 GiB
static inline bool windowed_slices(int variant) {
  return variant == kDefaultVariant || variant == kSmallGridVariant || variant == kX4D256Variant;   // the dispatched il kernels and the 256-wide kernel have a windowed instantiation
}

// Which instantiation of an il variant a launch runs: ONE rule for the launch switch (tfa_fwd_inst.inc), for TFA_FWD_EXACT_MAX's choice of variant 38
// (tfa_api.hip: pick_variant) and for tfa_fwd_rounding_rule — a (b,h) slice of 2 GiB and more takes the windowed form (il8 / il4 only), a single partial query
// block or packed GQA rows the idle-wave form, head dims that leave the kernel's last 32-column block empty the narrow form (il8 / il4 only), else the MAIN
// one: the instantiation with the hand-scheduled statement and, for bf16, the max-free row reference
enum IlInst { IL_MAIN = 0, IL_WINDOWED = 1, IL_IDLE = 2, IL_NARROW = 3 };
static inline IlInst il_instantiation(int variant, bool big, int Nq, int row_mod, int dv, int width) {
  if (variant == kExactVariant) return IL_MAIN;            // (pick_variant only hands it problems the main il8 instantiation would take)
  const bool has_special = variant == kDefaultVariant || variant == kSmallGridVariant;   // windowed / narrow forms exist
  const int bm = variant == kDefaultVariant ? 256 : 128;
  if (has_special && big) return IL_WINDOWED;
  if (Nq <= bm - 32 || row_mod) return IL_IDLE;
  if (has_special && dv <= width - 32) return IL_NARROW;
  return IL_MAIN;
}
static inline bool is_il_variant(int variant) {
  return variant == kDefaultVariant || variant == kSmallGridVariant || variant == kKSplitVariant || variant == kKSplitPairVariant || variant == kExactVariant;
}

static inline int block_m_of(int variant) {
  const Variant* v = variant_info(variant);
  if (!v) return 256;
  const int rows = v->nw * 32 * v->rb;
  return (v->vf & VF_IL_KSPLIT) ? rows / 2 : rows;   // two groups of waves share one query block
}
static inline bool pairs_causal(int variant) {
  const Variant* v = variant_info(variant);
  return v && (v->vf & VF_PAIR) != 0;
}

}  // namespace tfa
