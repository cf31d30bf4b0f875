// tfa_api.hip — the extern "C" boundary declared in include/tfa.h.
// Validates a problem descriptor, fills the kernel arguments, picks a kernel variant and
// launches on the caller's stream.  Mirrors what the reference host entry does
// (flash_attention_cutlass/csrc/flash_attention.cu:320-361 set_params_fprop, :731-739 dispatch,
//  :741-772 entry) minus allocation (the caller owns all buffers) and minus the device sync.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "tfa.h"
#include "tfa_launch.h"

namespace tfa {
// fp32 tensors (the reference's fp32 fixtures: a correctness path on v_mfma_f32_32x32x2_f32): tfa_fwd_f32.hip
hipError_t launch_f32(const KArgs& a, bool causal, hipStream_t stream, int* grid_out, bool dry);
template <> hipError_t launch_splitkv_wide<__bf16>(const KArgs&, bool, hipStream_t, LaunchGeom*, bool);
template <> hipError_t launch_splitkv_wide<_Float16>(const KArgs&, bool, hipStream_t, LaunchGeom*, bool);
}  // namespace tfa

namespace {

// Debug knobs (tfa_set_variant, tfa_debug_set_trace) are PER THREAD: a thread that forces a variant or a trace buffer
// for an A/B measurement does not change what any other thread's calls run.  Nothing else in the library is mutable.
thread_local int g_variant = -1;   // -1 = automatic
thread_local unsigned long long* g_trace = nullptr;
thread_local int g_dbg_flags = 0;                    // kernel bring-up flags (tfa_debug_set_flags): KArgs::dbg   // per-workgroup cycle stamps (tfa_debug_set_trace)

int num_cus() { return tfa::num_cus_current_device(); }

// every (b,h) slice of q, k, v and out fits ONE buffer descriptor (< 2 GiB including the rows a ragged block may reach past the
// end): what the key-split, split-KV, backward and x4 kernels need; larger slices run the windowed il4 / il8 instantiations
bool one_descriptor(const tfa_fwd_params* p) {
  const auto small = [&](int64_t n, const int64_t* st, int es) { return ((n + 512) * st[2] + p->D) * es < (int64_t)0x7fffffff; };
  return small(p->Nq, p->q_stride, 2) && small(p->Nk, p->k_stride, 2) && small(p->Nk, p->v_stride, 2) && small(p->Nq, p->o_stride, 4);
}

// An OUTPUT whose (b,h) slices share memory — a broadcast batch or head stride, or slices that interleave — would be written by several
// workgroups at once (inputs may broadcast freely: they are only read).  Row overlap inside a slice is the callers' `stride[2] < D` test.
bool out_aliases_itself(const tfa_fwd_params* p) {
  // the dims that have more than one index, by ascending stride: each stride must clear the span of everything below it (rows of D elements
  // at the bottom).  Sufficient, and every layout a tensor library hands out — (B,H,N,D), (B,N,H,D), slices and views of them — passes
  int64_t st[3] = {p->o_stride[0], p->o_stride[1], p->o_stride[2]};
  int64_t ext[3] = {p->B, p->H, p->Nq};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (st[j] < st[i]) { const int64_t t = st[i]; st[i] = st[j]; st[j] = t; const int64_t e = ext[i]; ext[i] = ext[j]; ext[j] = e; }
  int64_t span = p->D;
  for (int i = 0; i < 3; ++i) {
    if (ext[i] <= 1) continue;
    if (st[i] < span) return true;
    span = (ext[i] - 1) * st[i] + span;
  }
  return false;
}

// NOTE on tfa_set_variant (a per-thread debug knob): a forced variant means "run exactly that kernel on the problem as given" —
// GQA row packing (pack_gqa_rows) and tfa_fwd_suggest_splits are both switched off while one is set, and head dims above 128
// always run kX4D256Variant (the only kernel that wide).  Tools that force a variant reset it to -1 in a finally block.
int pick_variant(const tfa_fwd_params* p) {
  if (p && p->D > 128) return tfa::kX4D256Variant;   // one kernel serves 136..256 (a forced variant does not apply)
  if (g_variant >= 0) return g_variant;
  if (!p) return tfa::kDefaultVariant;
  if (p->flags & TFA_FWD_EXACT_MAX) {
    // the exact running max (validate: D <= 128): where the default would run il8 — grids that fill the chip, whole 256-row blocks' worth of
    // rows, slices below 2 GiB — its exact-max instantiation (round 5); everywhere else the burst-structured LDS-DMA kernel
    tfa_fwd_params d = *p;
    d.flags &= ~TFA_FWD_EXACT_MAX;
    // (variant 38 has the main instantiation only: the same rule the launch switch applies to variant 30 — tfa_launch.h: il_instantiation; packed rows never get here)
    const bool main30 = pick_variant(&d) == tfa::kDefaultVariant &&
                        tfa::il_instantiation(tfa::kDefaultVariant, !one_descriptor(p), p->Nq, 0, 128, 128) == tfa::IL_MAIN;
    return main30 ? tfa::kExactVariant : tfa::kSplitVariant;
  }
  // Measured on MI355X (tests/tools/ab.py, profiles/): 256-row query blocks (8 waves) are fastest when there
  // are enough of them to fill 256 CUs and no causal diagonal; 128-row blocks (two 4-wave workgroups per
  // CU) waste less of the causal diagonal and fill the chip on small problems (BASELINE config 2:
  // B4 H8 N1024 has only 128 blocks of 256 rows).
  // The issue-interleaved kernel (tfa_fwd_kernel_il.h) wins wherever the grid fills the chip (+5..11% at D=128).
  const long long blocks256 = (long long)p->B * p->H * ((p->Nq + 255) / 256);
  const long long blocks128 = (long long)p->B * p->H * ((p->Nq + 127) / 128);
  const int cus = num_cus();
  // (partial passes — kv_offset / nk_total — reach the kernels as a causal shift only: every rule below applies to them too)
  // Grids of at most one 128-row block per CU: the 4-wave kernel would run one wave per SIMD (causal: paired, on half the CUs) —
  // split the keys inside the workgroup instead (il8-ksplit: 8 waves on one block, unpaired).  Causal it always pays (B1 H8 N4096 +15 %,
  // B1 H16 N2048 +19 %, B1 H64 N512 +27 %, B4 H8 N1024 D64 +31 % over il4).  Non-causal the 4-wave kernel with the hand-scheduled tile loop
  // (round 5: +8..19 % on these grids; the key-split waves see half the tiles each and gain 0..3 %) is ahead up to 2048 keys (BASELINE config 2:
  // 611 vs 542 TF, B4 H8 N1024 D128 816 vs 762, B1 H16 N2048 955 vs 931) and level at 4096 (1043 vs 1056): key split from 4096 keys on
  // (profiles/r05_ksplit_retune.txt; rounds 2-4 had it from 1024: profiles/r02_ksplit_ab.txt).
  const bool one_desc = one_descriptor(p);   // (slices of 2 GiB and more: the windowed il4 / il8 instantiations)
  if (one_desc && blocks128 <= cus && p->Nk >= (p->is_causal ? 512 : 4096)) return tfa::kKSplitVariant;
  // causal, up to two 128-row blocks per CU, long sequences: the same kernel with the blocks paired heavy+light (one round of
  // equal workgroups, two waves per SIMD): B1 H16 N4096 +4 %, B1 H8 N8192 +7 %, B1 H4 N16384 +11 % over il4; N=2048: -2..+5 %
  if (one_desc && p->is_causal && blocks128 <= 2 * cus && p->Nk >= 4096) return tfa::kKSplitPairVariant;
  // at most 128 query rows (decode, cross-attention onto few queries): a 256-row block would be half idle; 128-row blocks put two
  // workgroups on a CU and keep twice the K/V bytes in flight (B32 H32 Nq1 Nk16384 D64: K/V at 6.1 vs 5.0 TB/s, D128: 6.1 vs 6.0)
  if (p->Nq <= 128) return tfa::kSmallGridVariant;
  // non-causal, at least one 256-row block per CU: the 8-wave kernel already has two waves per SIMD everywhere
  // (B1 H16 N4096: 1160 vs 1091 TF for il4, B1 H32 N2048: 1098 vs 1038)
  if (!p->is_causal && blocks256 >= cus) return tfa::kDefaultVariant;
  // small grids: 128-row blocks, two 4-wave workgroups per CU, issue-interleaved, O through the idle tile buffers
  // (D=128: +8..15 % over the burst kernel on B1 H8 N2048 / B2 H16 N1024 / B1 H32 N4096; D=64, BASELINE config 2: +2 %)
  if (blocks256 < 512) return tfa::kSmallGridVariant;
  // (with O leaving through LDS the 8-wave il kernel also wins on short sequences: N=512..2048, causal or not, it beats
  //  the 4-wave one by 3-5 %, tests/tools/ab.py n512/n1k/n2k configs)
  return tfa::kDefaultVariant;
}

// extent in bytes of one (b,h) slice: rows 0..N-1 at row stride, D contiguous elements each.  Kernels with one descriptor
// per slice need every byte offset they form (up to one block past the end) inside int32; the il / x4 kernels address the
// slice through per-block / per-tile windows (rsrc_at), so only a WINDOW (a 256-row query block or a 64-row tile plus the
// rows a ragged tail may reach past it) has to fit 2 GiB — long (B,N,H,D) tensors whose head slices span more are fine.
bool slice_bytes(int64_t n, int64_t row_stride, int d, int esize, bool windowed, unsigned long long* out, int* big) {
  const int64_t bytes = ((n - 1) * row_stride + d) * esize;
  const int64_t whole = ((n + 512) * row_stride + d) * esize;       // one descriptor per slice
  const int64_t window = (768 * row_stride + d) * esize;            // one per query block / tile
  if (bytes <= 0) return false;
  if (whole >= (int64_t)0x7fffffff) {
    if (!windowed || window >= (int64_t)0x7fffffff) return false;
    *big = 1;
  }
  *out = (unsigned long long)bytes;
  return true;
}

int validate(const tfa_fwd_params* p, tfa::KArgs* a, int variant, int row_mod = 0, bool wide_split = false) {
  if (!p) return TFA_ERR_NULL;
  if (!p->q || !p->k || !p->v || !p->out) return TFA_ERR_NULL;
  if (p->dtype != TFA_F16 && p->dtype != TFA_BF16) return TFA_ERR_DTYPE;
  if (p->out_dtype != p->dtype && p->out_dtype != TFA_F32) return TFA_ERR_DTYPE;
  if (p->D < 8 || p->D > 256 || (p->D % 8) != 0) return TFA_ERR_HEAD_DIM;
  if (p->B <= 0 || p->H <= 0 || p->Hk <= 0 || p->Nq <= 0 || p->Nk <= 0) return TFA_ERR_SHAPE;
  if (p->H % p->Hk != 0) return TFA_ERR_SHAPE;
  if (!(p->softmax_scale > 0.f) || !isfinite(p->softmax_scale)) return TFA_ERR_SCALE;
  if ((p->flags & ~TFA_FWD_EXACT_MAX) != 0 || p->reserved_ != 0) return TFA_ERR_SHAPE;
  if ((p->flags & TFA_FWD_EXACT_MAX) && p->D > 128) return TFA_ERR_HEAD_DIM;          // no exact-max kernel that wide
  const bool ablate = (variant >= 100 && variant < 100 + 512) || (variant >= 700 && variant < 716) || (variant >= 1000 && variant < 2256) || (variant >= 3000 && variant < 3256);   // timing-only ablations (debug)
  if (!ablate && !tfa::variant_built(variant)) return TFA_ERR_VARIANT;
  if (p->D != 64 && p->D != 128 && (ablate || !tfa::supports_padded_d(variant))) return TFA_ERR_HEAD_DIM;   // (A/B arms: 64 / 128 only)
  // (wide_split: tfa_fwd_splitkv's partial pass — the LDS-DMA kernel exists 256 wide for that purpose only)
  if (p->D > 128 ? !(variant == tfa::kX4D256Variant || (wide_split && variant == tfa::kSplitVariant)) : variant == tfa::kX4D256Variant) return TFA_ERR_HEAD_DIM;
  const int esz = 2, osz = (p->out_dtype == TFA_F32) ? 4 : 2;
  const int64_t* st[4] = {p->q_stride, p->k_stride, p->v_stride, p->o_stride};
  for (int t = 0; t < 4; ++t) {
    const int es = (t == 3) ? osz : esz;
    for (int i = 0; i < 3; ++i) {
      if (st[t][i] < 0) return TFA_ERR_STRIDE;
      if ((st[t][i] * es) % 16 != 0) return TFA_ERR_STRIDE;   // 16-byte vector access on every row
    }
    if (st[t][2] < p->D) return TFA_ERR_STRIDE;               // rows must not overlap
  }
  if (out_aliases_itself(p)) return TFA_ERR_STRIDE;
  if (((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->out) & 15) return TFA_ERR_ALIGN;
  if (p->lse && ((uintptr_t)p->lse & 3)) return TFA_ERR_ALIGN;

  memset(a, 0, sizeof(*a));
  a->q = p->q; a->k = p->k; a->v = p->v; a->o = p->out; a->lse = p->lse;
  a->B = p->B; a->H = p->H; a->Hk = p->Hk; a->Nq = p->Nq; a->Nk = p->Nk;
  {
    if (p->kv_offset < 0 || p->nk_total < 0) return TFA_ERR_SHAPE;
    const int64_t total = p->nk_total ? p->nk_total : (p->kv_offset + p->Nk);
    if (p->kv_offset + p->Nk > total || total - p->Nq - p->kv_offset < -(int64_t)0x3fffffff || total >= (int64_t)0x3fffffff) return TFA_ERR_SHAPE;
    a->shift = (int)(total - p->Nq - p->kv_offset);
  }
  a->qs_b = p->q_stride[0]; a->qs_h = p->q_stride[1]; a->qs_n = p->q_stride[2];
  a->ks_b = p->k_stride[0]; a->ks_h = p->k_stride[1]; a->ks_n = p->k_stride[2];
  a->vs_b = p->v_stride[0]; a->vs_h = p->v_stride[1]; a->vs_n = p->v_stride[2];
  a->os_b = p->o_stride[0]; a->os_h = p->o_stride[1]; a->os_n = p->o_stride[2];
  const bool win = !ablate && tfa::windowed_slices(variant);
  if (!slice_bytes(p->Nq, a->qs_n, p->D, esz, win, &a->q_bytes, &a->big)) return TFA_ERR_STRIDE;
  if (!slice_bytes(p->Nk, a->ks_n, p->D, esz, win, &a->k_bytes, &a->big)) return TFA_ERR_STRIDE;
  if (!slice_bytes(p->Nk, a->vs_n, p->D, esz, win, &a->v_bytes, &a->big)) return TFA_ERR_STRIDE;
  if (!slice_bytes(p->Nq, a->os_n, p->D, osz, win, &a->o_bytes, &a->big)) return TFA_ERR_STRIDE;
  if ((g_dbg_flags & 256) && win) a->big = 1;   // tests: run the windowed instantiation on a small problem
  a->scale = p->softmax_scale;
  a->trace = g_trace;
  a->grid = num_cus();
  if (g_dbg_flags & 512) a->grid = 8;            // tests: persistent kernels with 8 workgroups, so that small problems walk several work items each
  a->scale_log2 = p->softmax_scale * 1.4426950408889634f;
  const int bm = ablate ? 256 : tfa::block_m_of(variant);
  a->nmb = (p->Nq + bm - 1) / bm;
  a->nwork = (p->is_causal && (ablate || tfa::pairs_causal(variant))) ? (a->nmb + 1) / 2 : a->nmb;
  const int64_t nbh = (int64_t)p->B * p->H;
  if (nbh * a->nwork >= (int64_t)0x7fffffff) return TFA_ERR_SHAPE;
  a->nbh = (int)nbh;
  a->dbg = g_dbg_flags;
  // K and V together beyond 768 MiB: the cache will not survive in the memory-side cache until the next call -> decode kernels may
  // stream it with the non-temporal hint (tfa_fwd_kernel_il.h / tfa_fwd_kernel_dma.h: kv_private, VF_DMA_NT)
  a->kv_stream = ((long long)p->B * p->Hk * p->Nk * p->D * 4 >= (768ll << 20)) ? 1 : 0;
  a->row_mod = row_mod;
  a->dv = p->D;
  return TFA_OK;
}

// GQA / MQA with few query rows (decode): the G = H/Hk query heads that share a K/V head are G x Nq ROWS of one problem over
// that head's keys — the same bytes described differently (head stride G times larger, rows one head apart), so K and V are
// streamed once per K/V head instead of once per query head and the grid shrinks G-fold.  One query row: any strides, and
// a causal mask hides nothing from it (its position is the last key), so the packed problem is non-causal.  More rows: only
// non-causal and with the heads of q / out adjacent in memory (rows of consecutive heads are then equidistant).  The LSE layout
// (B,H,Nq) is the packed problem's (B,Hk,G*Nq) as it stands.  Causal with several rows (speculative decoding, the tail of a
// chunked prefill): row r of the packed block is query position r % Nq, which the decode instantiations of the il kernels
// understand (KArgs::row_mod, *row_mod here; causal_rows says whether the caller's kernel is one of them); nk_total carries
// the causal shift of the ORIGINAL problem (Nk - Nq) past the larger row count.  Returns false when *p is not such a problem.
bool pack_gqa_rows(const tfa_fwd_params* p, tfa_fwd_params* o, int* row_mod = nullptr, bool causal_rows = false) {
  if (row_mod) *row_mod = 0;
  if (!p || p->Hk <= 0 || p->H <= p->Hk || p->H % p->Hk != 0 || p->Nq <= 0 || p->Nk <= 0) return false;
  if (p->kv_offset != 0 || p->nk_total != 0) return false;
  const int G = p->H / p->Hk;
  const bool one_row = p->Nq == 1;
  const bool adjacent = p->q_stride[1] == (int64_t)p->Nq * p->q_stride[2] && p->o_stride[1] == (int64_t)p->Nq * p->o_stride[2];
  const bool with_positions = !one_row && p->is_causal;
  if (!one_row && !adjacent) return false;
  if (with_positions && !(causal_rows && row_mod && p->D <= 128 && p->Nk >= p->Nq)) return false;
  if ((long long)G * p->Nq > 128) return false;          // beyond one query block nothing is shared any more
  *o = *p;
  o->H = p->Hk;
  o->Nq = G * p->Nq;
  if (one_row) {
    o->q_stride[2] = p->q_stride[1];
    o->o_stride[2] = p->o_stride[1];
    o->is_causal = 0;
  }
  if (with_positions) {
    *row_mod = p->Nq;
    o->nk_total = (int64_t)p->Nk + (int64_t)(G - 1) * p->Nq;   // shift = nk_total - G*Nq = Nk - Nq
  }
  o->q_stride[1] = (int64_t)G * p->q_stride[1];
  o->o_stride[1] = (int64_t)G * p->o_stride[1];
  return true;
}

// fp32 q, k, v (tfa_fwd_params::dtype == TFA_F32): the correctness path behind the reference's fp32 fixtures (tfa_fwd_f32.hip).  fp32 output
// only; any strides with 16-byte aligned rows; head dims = multiples of 4 up to 256; GQA, Nq != Nk, kv_offset / nk_total as for the 16-bit types.
int run_f32(const tfa_fwd_params* p, void* stream, tfa::LaunchGeom* geom, bool dry) {
  if (!p->q || !p->k || !p->v || !p->out) return TFA_ERR_NULL;
  if (p->out_dtype != TFA_F32) return TFA_ERR_DTYPE;
  if (p->D < 4 || p->D > 256 || (p->D % 4) != 0) return TFA_ERR_HEAD_DIM;
  if (p->B <= 0 || p->H <= 0 || p->Hk <= 0 || p->Nq <= 0 || p->Nk <= 0 || p->H % p->Hk != 0) return TFA_ERR_SHAPE;
  if (!(p->softmax_scale > 0.f) || !isfinite(p->softmax_scale)) return TFA_ERR_SCALE;
  if (p->flags != 0 || p->reserved_ != 0) return TFA_ERR_SHAPE;          // (TFA_FWD_EXACT_MAX is about 16-bit rounding points: this path has none)
  const int64_t* st[4] = {p->q_stride, p->k_stride, p->v_stride, p->o_stride};
  for (int t = 0; t < 4; ++t) {
    for (int i = 0; i < 3; ++i)
      if (st[t][i] < 0 || (st[t][i] * 4) % 16 != 0) return TFA_ERR_STRIDE;
    if (st[t][2] < p->D) return TFA_ERR_STRIDE;
  }
  if (out_aliases_itself(p)) return TFA_ERR_STRIDE;
  if (p->Nq >= 0x3fffffff) return TFA_ERR_SHAPE;                          // (the kernel forms row + 32 + shift in 32-bit arithmetic)
  if (((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->out) & 15) return TFA_ERR_ALIGN;
  if (p->lse && ((uintptr_t)p->lse & 3)) return TFA_ERR_ALIGN;
  tfa::KArgs a;
  memset(&a, 0, sizeof(a));
  a.q = p->q; a.k = p->k; a.v = p->v; a.o = p->out; a.lse = p->lse;
  a.B = p->B; a.H = p->H; a.Hk = p->Hk; a.Nq = p->Nq; a.Nk = p->Nk;
  if (p->kv_offset < 0 || p->nk_total < 0) return TFA_ERR_SHAPE;
  const int64_t total = p->nk_total ? p->nk_total : (p->kv_offset + p->Nk);
  if (p->kv_offset + p->Nk > total || total - p->Nq - p->kv_offset < -(int64_t)0x3fffffff || total >= (int64_t)0x3fffffff) return TFA_ERR_SHAPE;
  a.shift = (int)(total - p->Nq - p->kv_offset);
  a.qs_b = p->q_stride[0]; a.qs_h = p->q_stride[1]; a.qs_n = p->q_stride[2];
  a.ks_b = p->k_stride[0]; a.ks_h = p->k_stride[1]; a.ks_n = p->k_stride[2];
  a.vs_b = p->v_stride[0]; a.vs_h = p->v_stride[1]; a.vs_n = p->v_stride[2];
  a.os_b = p->o_stride[0]; a.os_h = p->o_stride[1]; a.os_n = p->o_stride[2];
  a.scale = p->softmax_scale;
  a.scale_log2 = p->softmax_scale * 1.4426950408889634f;
  const int64_t nbh = (int64_t)p->B * p->H;
  if (nbh * ((p->Nq + 31) / 32) >= (int64_t)0x7fffffff) return TFA_ERR_SHAPE;
  a.nbh = (int)nbh;
  a.dv = p->D;
  int grid = 0;
  const hipError_t e = tfa::launch_f32(a, p->is_causal != 0, reinterpret_cast<hipStream_t>(stream), &grid, dry);
  if (geom) { geom->grid = grid; geom->block = 256; geom->lds = 0; }
  return (int)e;
}

int run(const tfa_fwd_params* p_in, void* stream, tfa::LaunchGeom* geom, bool dry, int* variant_out = nullptr, int* rule_out = nullptr) {
  if (p_in && p_in->dtype == TFA_F32) {
    if (variant_out) *variant_out = -1;
    if (rule_out) *rule_out = TFA_RULE_EXACT_MAX;
    return run_f32(p_in, stream, geom, dry);
  }
  tfa_fwd_params packed;
  int row_mod = 0;
  const bool may_pack = g_variant < 0 && !(g_dbg_flags & 4096) && p_in && !(p_in->flags & TFA_FWD_EXACT_MAX);
  const tfa_fwd_params* p = (may_pack && pack_gqa_rows(p_in, &packed, &row_mod, true)) ? &packed : p_in;
  int variant = pick_variant(p);
  tfa::KArgs a;
  int st = validate(p, &a, variant, row_mod);
  // The packed description is an optimisation, never a requirement: whenever it does not validate (packing moves the head
  // stride into the row-stride slot, so e.g. a q broadcast over heads — head stride 0 — fails the "rows do not overlap" test)
  // or needs the windowed instantiations (which know nothing of packed positions), the problem runs as the caller gave it.
  if (p != p_in && (st != TFA_OK || (a.big && row_mod))) {
    p = p_in;
    row_mod = 0;
    variant = pick_variant(p);
    st = validate(p, &a, variant);
  }
  if (st != TFA_OK) return st;
  if (variant_out) *variant_out = variant;
  if (rule_out) {
    // the row reference P is rounded against (include/tfa.h): the il kernels' lazily re-based reference, except — bf16, the main instantiation (the one
    // that carries the hand-scheduled statement: tfa_fwd_kernel_il.h MAXFREE) — the first key tile's maximum; the 256-wide kernel re-bases lazily too
    const tfa::Variant* vi = tfa::variant_info(variant);
    const bool il = vi && (vi->vf & tfa::VF_IL) != 0 && variant != tfa::kExactVariant;
    const bool x4 = p->D > 128;
    const bool main_inst = tfa::is_il_variant(variant) && tfa::il_instantiation(variant, a.big != 0, a.Nq, a.row_mod, a.dv, p->D > 64 ? 128 : 64) == tfa::IL_MAIN;
    *rule_out = x4 ? TFA_RULE_LAZY : !il ? TFA_RULE_EXACT_MAX : (p->dtype == TFA_BF16 && main_inst && TFA_IL_USE_MAXFREE) ? TFA_RULE_FIRST_TILE : TFA_RULE_LAZY;
  }
  const bool causal = p->is_causal != 0;
  const bool f32out = p->out_dtype == TFA_F32;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e;
  const bool wide = p->D > 64;   // kernel width: 64 serves D <= 64, 128 serves 64 < D <= 128, 256 the rest (KArgs::dv = the valid part)
  if (p->D > 128) {
    e = (p->dtype == TFA_BF16) ? tfa::launch_x4_unit<__bf16, 256>(a, causal, f32out, 0, s, geom, dry)
                               : tfa::launch_x4_unit<_Float16, 256>(a, causal, f32out, 0, s, geom, dry);
  } else if (p->dtype == TFA_BF16) {
    e = wide ? tfa::launch_fwd<__bf16, 128>(a, causal, f32out, variant, s, geom, dry)
             : tfa::launch_fwd<__bf16, 64>(a, causal, f32out, variant, s, geom, dry);
  } else {
    e = wide ? tfa::launch_fwd<_Float16, 128>(a, causal, f32out, variant, s, geom, dry)
             : tfa::launch_fwd<_Float16, 64>(a, causal, f32out, variant, s, geom, dry);
  }
  return (int)e;
}

void fill_bhnd(tfa_fwd_params* p, const void* q, const void* k, const void* v, void* out, float* lse,
               int B, int H, int N, int D, float scale, int causal, int dtype, int out_dtype) {
  memset(p, 0, sizeof(*p));
  p->q = q; p->k = k; p->v = v; p->out = out; p->lse = lse;
  p->B = B; p->H = H; p->Hk = H; p->Nq = N; p->Nk = N; p->D = D;
  const int64_t sb = (int64_t)H * N * D, sh = (int64_t)N * D, sn = D;
  int64_t* st[4] = {p->q_stride, p->k_stride, p->v_stride, p->o_stride};
  for (int t = 0; t < 4; ++t) { st[t][0] = sb; st[t][1] = sh; st[t][2] = sn; }
  p->softmax_scale = scale; p->is_causal = causal; p->dtype = dtype; p->out_dtype = out_dtype;
}

}  // namespace

extern "C" {

int tfa_version(void) { return TFA_VERSION; }

const char* tfa_strerror(int status) {
  switch (status) {
    case TFA_OK: return "success";
    case TFA_ERR_NULL: return "tfa: a required pointer is NULL";
    case TFA_ERR_DTYPE: return "tfa: unsupported dtype (q/k/v must be fp16 or bf16, out matching or fp32; or q/k/v fp32 with fp32 out: the correctness path)";
    case TFA_ERR_HEAD_DIM: return "tfa: unsupported head dim (forward, split-KV and backward: multiples of 8 up to 256; TFA_FWD_EXACT_MAX: multiples of 8 up to 128; merge: multiples of 4 up to 256)";
    case TFA_ERR_SHAPE: return "tfa: bad shape (sizes must be positive and H % Hk == 0)";
    case TFA_ERR_STRIDE: return "tfa: bad stride (must be >=0, rows 16-byte aligned and non-overlapping; 768 rows of a (b,h) slice must span < 2 GiB, the whole slice for split-KV / backward)";
    case TFA_ERR_ALIGN: return "tfa: base pointers must be 16-byte aligned";
    case TFA_ERR_VARIANT: return "tfa: unknown kernel variant";
    case TFA_ERR_SCALE: return "tfa: softmax_scale must be finite and > 0";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "tfa: unknown status";
}

int tfa_fwd(const tfa_fwd_params* p, void* stream) { return run(p, stream, nullptr, false); }

int tfa_fwd_bhnd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int N,
                 int D, float softmax_scale, int is_causal, int dtype, void* stream) {
  tfa_fwd_params p;
  fill_bhnd(&p, q, k, v, out, lse, B, H, N, D, softmax_scale, is_causal, dtype, dtype);
  return run(&p, stream, nullptr, false);
}

int tfa_fwd_bhnd_f32out(const void* q, const void* k, const void* v, float* out, float* lse, int B, int H,
                        int N, int D, float softmax_scale, int is_causal, int dtype, void* stream) {
  tfa_fwd_params p;
  fill_bhnd(&p, q, k, v, out, lse, B, H, N, D, softmax_scale, is_causal, dtype, TFA_F32);
  return run(&p, stream, nullptr, false);
}

int tfa_fwd_plan(const tfa_fwd_params* p, int* grid, int* block, int* lds_bytes) {
  tfa::LaunchGeom g = {0, 0, 0};
  const int st = run(p, nullptr, &g, true);
  if (st != 0) return st;
  if (grid) *grid = g.grid;
  if (block) *block = g.block;
  if (lds_bytes) *lds_bytes = g.lds;
  return TFA_OK;
}

// ---- side streams for the one-launch-per-chunk route of tfa_fwd_splitkv ----------------------------------------------------
// The chunk launches of one call are independent; on ONE stream they run one after the other and a decode-like problem (few
// workgroups per launch) fills the chip no better than tfa_fwd.  They are therefore forked over a few side streams and joined
// back into the caller's stream before the merge (event fork / join: legal inside a stream capture too).  The streams and events
// belong to the calling thread and the current device; they are created on first use and live as long as the thread.
constexpr int kSideStreams = 4;
struct SideStreams {
  int device = -1;
  hipStream_t s[kSideStreams] = {};
  hipEvent_t fork = nullptr, join[kSideStreams] = {};
  bool ok = false;
};
SideStreams* side_streams() {
  thread_local SideStreams pools[8];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  SideStreams* sp = nullptr;
  for (auto& q : pools)
    if (q.device == dev) { sp = &q; break; }
  if (!sp)
    for (auto& q : pools)
      if (q.device < 0) { sp = &q; break; }
  if (!sp) return nullptr;                                   // (more than eight devices driven by one thread: run the chunks in line)
  if (sp->device == dev) return sp->ok ? sp : nullptr;
  sp->device = dev;
  bool ok = hipEventCreateWithFlags(&sp->fork, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < kSideStreams && ok; ++i)
    ok = hipStreamCreateWithFlags(&sp->s[i], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&sp->join[i], hipEventDisableTiming) == hipSuccess;
  sp->ok = ok;
  if (!ok) (void)hipGetLastError();
  return ok ? sp : nullptr;
}

// ---- split-KV in one launch ------------------------------------------------------------------------------------------
static int splitkv_geometry(const tfa_fwd_params* p, int splits, int* nsplit, int* chunk) {
  if (!p || splits < 1) return TFA_ERR_SHAPE;
  if (p->dtype == TFA_F32) return TFA_ERR_DTYPE;                             // fp32 q, k, v: tfa_fwd only (tfa.h) — whichever route the call would take
  if (p->kv_offset != 0 || p->nk_total != 0) return TFA_ERR_SHAPE;          // the call splits the WHOLE key sequence
  int c = (p->Nk + splits - 1) / splits;
  c = (c + 63) / 64 * 64;
  *chunk = c;
  *nsplit = (p->Nk + c - 1) / c;
  return TFA_OK;
}

long long tfa_fwd_splitkv_workspace(const tfa_fwd_params* p, int splits) {
  int ns = 0, ch = 0;
  const int st = splitkv_geometry(p, splits, &ns, &ch);
  if (st != TFA_OK) return st;
  const long long rows = (long long)p->B * p->H * p->Nq;
  return (long long)ns * rows * (p->D + 1);
}

int tfa_fwd_splitkv(const tfa_fwd_params* p, int splits, float* workspace, void* stream) {
  int ns = 0, ch = 0;
  int st = splitkv_geometry(p, splits, &ns, &ch);
  if (st != TFA_OK) return st;
  if (p->D < 8 || p->D > 256 || (p->D % 8) != 0) return TFA_ERR_HEAD_DIM;
  if (p->flags & TFA_FWD_EXACT_MAX) return TFA_ERR_SHAPE;                    // (the merge moves the rounding points: see tfa.h)
  if (!workspace || ((uintptr_t)workspace & 15)) return workspace ? TFA_ERR_ALIGN : TFA_ERR_NULL;
  // the merge writes contiguous rows: out must be a contiguous (B,H,Nq,D) tensor
  if (p->o_stride[2] != p->D || p->o_stride[1] != (int64_t)p->Nq * p->D || p->o_stride[0] != (int64_t)p->H * p->Nq * p->D) return TFA_ERR_STRIDE;
  const long long rows = (long long)p->B * p->H * p->Nq;
  float* ws_o = workspace;
  float* ws_l = workspace + (long long)ns * rows * p->D;
  tfa_fwd_params q = *p;                                  // the partial pass: fp32 O and LSE of every chunk into the workspace
  q.out = ws_o;
  q.lse = ws_l;
  q.out_dtype = TFA_F32;
  q.o_stride[0] = (int64_t)p->H * p->Nq * p->D; q.o_stride[1] = (int64_t)p->Nq * p->D; q.o_stride[2] = p->D;
  if (!one_descriptor(p) || (g_dbg_flags & 8192)) {
    // (b,h) slices of 2 GiB and more (long strided K/V caches): the LDS-DMA kernel below addresses a slice through ONE descriptor,
    // tfa_fwd's kernels through windows — the partial passes are `ns` launches of tfa_fwd over key chunks (kv_offset / nk_total:
    // the causal mask stays against global key positions), one launch per chunk instead of one in all, same partials, same
    // merge.  (Debug flag 8192 forces this route: tests compare it with the one-launch form.)
    const int64_t rs_k = p->k_stride[2], rs_v = p->v_stride[2];
    hipStream_t caller = reinterpret_cast<hipStream_t>(stream);
    // chunks that leave the chip mostly idle run side by side on the thread's side streams (debug flag 16384: in line)
    const long long blocks = (long long)p->B * p->H * ((p->Nq + 127) / 128);
    SideStreams* const pool = (ns >= 2 && blocks * 2 <= num_cus() && !(g_dbg_flags & 16384)) ? side_streams() : nullptr;
    SideStreams* ss = pool;
    // fork: every side stream waits for the caller's stream.  A stream that has been forked MUST be joined back before this function
    // returns, whatever happens in between (inside a stream capture an unjoined fork invalidates the capture): `forked` counts them,
    // and a failure half way through the fork simply runs the chunks in line on the caller's stream.
    int forked = 0;
    if (ss) {
      if (hipEventRecord(ss->fork, caller) == hipSuccess) {
        for (; forked < kSideStreams; ++forked)
          if (hipStreamWaitEvent(ss->s[forked], ss->fork, 0) != hipSuccess) break;
      }
      if (forked < kSideStreams) { (void)hipGetLastError(); ss = nullptr; }   // partial fork: joined below, chunks in line
    }
    int st_chunks = TFA_OK;
    for (int c = 0; c < ns && st_chunks == TFA_OK; ++c) {
      tfa_fwd_params qc = q;
      const int64_t k0 = (int64_t)c * ch;
      qc.k = reinterpret_cast<const char*>(p->k) + k0 * rs_k * 2;
      qc.v = reinterpret_cast<const char*>(p->v) + k0 * rs_v * 2;
      qc.Nk = (int)((p->Nk - k0) < ch ? (p->Nk - k0) : ch);
      qc.kv_offset = k0;
      qc.nk_total = p->Nk;
      qc.out = ws_o + (long long)c * rows * p->D;
      qc.lse = ws_l + (long long)c * rows;
      st_chunks = run(&qc, ss ? (void*)ss->s[c % kSideStreams] : stream, nullptr, false);
    }
    // join — every forked stream, also after a failed launch or a partial fork: the caller's stream must not lose the fork
    int st_join = TFA_OK;
    if (forked > 0) {
      for (int i = 0; i < forked; ++i) {                   // (`pool`, not `ss`: a partial fork dropped ss above)
        hipError_t e = hipEventRecord(pool->join[i], pool->s[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(caller, pool->join[i], 0);
        if (e != hipSuccess && st_join == TFA_OK) { st_join = (int)e; (void)hipGetLastError(); }   // keep joining the others
      }
    }
    if (st_chunks != TFA_OK) return st_chunks;
    if (st_join != TFA_OK) return st_join;                 // (HIP errors are reported as positive status codes: tfa_strerror)
    return tfa_merge(ws_o, ws_l, ns, rows, p->D, rows * p->D, rows, p->out, p->out_dtype, p->lse, stream);
  }
  const int variant = tfa::kSplitVariant;                 // the LDS-DMA kernel carries the chunk dimension in its grid
  tfa::KArgs a;
  {
    // GQA decode: one stream of K/V per K/V head (the workspace rows keep their order).  As in run(): the packed description is an
    // optimisation, never a requirement — when it does not validate (a q broadcast over heads: head stride 0; a row stride that
    // pushes (G + 512) rows past 2 GiB) the problem runs as the caller gave it.
    tfa_fwd_params qp;
    st = TFA_ERR_SHAPE;
    if (!(g_dbg_flags & 4096) && pack_gqa_rows(&q, &qp)) {
      st = validate(&qp, &a, variant, 0, true);
      if (st == TFA_OK) q = qp;
    }
    if (st != TFA_OK) st = validate(&q, &a, variant, 0, true);
  }
  if (st != TFA_OK) return st;
  a.nsplit = ns;
  a.chunk = ch;

  a.o_part_stride = rows * p->D;
  a.lse_part_stride = rows;
  if ((long long)a.nbh * a.nwork * ns >= (long long)0x7fffffff) return TFA_ERR_SHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e;
  const bool causal = q.is_causal != 0;                   // (the packed one-row problem is non-causal)
  if (p->D > 128)                                          // head dims 136..256: the same kernel 256 wide (one wave per SIMD, hand-owned accumulators)
    e = (p->dtype == TFA_BF16) ? tfa::launch_splitkv_wide<__bf16>(a, causal, s, nullptr, false) : tfa::launch_splitkv_wide<_Float16>(a, causal, s, nullptr, false);
  else if (p->dtype == TFA_BF16)
    e = (p->D > 64) ? tfa::launch_fwd<__bf16, 128>(a, causal, true, variant, s, nullptr, false) : tfa::launch_fwd<__bf16, 64>(a, causal, true, variant, s, nullptr, false);
  else
    e = (p->D > 64) ? tfa::launch_fwd<_Float16, 128>(a, causal, true, variant, s, nullptr, false) : tfa::launch_fwd<_Float16, 64>(a, causal, true, variant, s, nullptr, false);
  if (e != hipSuccess) return (int)e;
  return tfa_merge(ws_o, ws_l, ns, rows, p->D, rows * p->D, rows, p->out, p->out_dtype, p->lse, stream);
}

int tfa_fwd_suggest_splits(const tfa_fwd_params* p_in) {
  if (!p_in || g_variant >= 0) return 1;                  // a forced kernel variant means: run exactly that
  if (p_in->dtype == TFA_F32) return 1;                   // (the fp32 correctness path is one pass)
  if (p_in->flags & TFA_FWD_EXACT_MAX) return 1;          // (the merge of partial passes moves the rounding points as well)
  tfa_fwd_params packed;
  const tfa_fwd_params* p = pack_gqa_rows(p_in, &packed) ? &packed : p_in;
  // ((b,h) slices of 2 GiB and more take tfa_fwd_splitkv's one-launch-per-chunk route, the launches spread over side streams: a
  //  small chunk count, below)
  if (p->kv_offset != 0 || p->nk_total != 0 || p->B <= 0 || p->H <= 0 || p->Nq <= 0) return 1;
  const long long blocks = (long long)p->B * p->H * ((p->Nq + 127) / 128);
  const int cus = num_cus();
  if (blocks * 2 > cus || p->Nk < 4096) return 1;
  if (p->is_causal && (long long)p->Nq * 4 > p->Nk) return 1;   // causal prefill: the late chunks serve few rows (measured 0.93-1.06x)
  // up to a quarter of the CUs: one chunk per idle CU (measured 3-9x).  Between a quarter and a half: the split kernel fits two
  // workgroups per CU, fill those (blocks 96: 226 -> 141 us with 4 chunks, 128: 230-240 -> 165-175 us; at 160-192 blocks a split
  // no longer pays: profiles/r03_decode_nt_ab.txt)
  long long s = blocks * 4 > cus ? 2 * cus / blocks : cus / blocks;
  if (s > p->Nk / 1024) s = p->Nk / 1024;
  if (s > 32) s = 32;
  // one launch per chunk (slices of 2 GiB and more): every chunk costs a launch on the host and the four side
  // streams overlap about two launches' worth — measured 1.4-1.8x over one pass at four chunks, less at eight or sixteen
  // (tools/bench_decode_wide.py, profiles/r03_decode_wide.txt)
  if (!one_descriptor(p_in) && s > 4) s = 4;            // (the caller's strides, as tfa_fwd_splitkv tests them — not the packed ones)
  return s >= 2 ? (int)s : 1;
}

int tfa_fwd_variant(const tfa_fwd_params* p) {
  tfa::LaunchGeom g = {0, 0, 0};
  int variant = -1;                                       // run()'s final choice (after GQA packing and its fall-back)
  const int st = run(p, nullptr, &g, true, &variant);
  if (st != 0) return st > 0 ? TFA_ERR_SHAPE : st;
  return variant;
}

int tfa_fwd_rounding_rule(const tfa_fwd_params* p) {
  tfa::LaunchGeom g = {0, 0, 0};
  int variant = -1, rule = -1;
  const int st = run(p, nullptr, &g, true, &variant, &rule);
  if (st != 0) return st > 0 ? TFA_ERR_SHAPE : st;
  return rule;
}

int tfa_fwd_time(const tfa_fwd_params* p, int warmup, int iters, void* stream, float* avg_ms) {
  if (!avg_ms || iters <= 0 || warmup < 0) return TFA_ERR_NULL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < warmup; ++i) {
    const int st = run(p, stream, nullptr, false);
    if (st != 0) return st;
  }
  hipEvent_t e0, e1;
  hipError_t e = hipEventCreate(&e0);
  if (e != hipSuccess) return (int)e;
  e = hipEventCreate(&e1);
  if (e != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
  int st = 0;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && st == 0; ++i) st = run(p, stream, nullptr, false);
  (void)hipEventRecord(e1, s);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (st != 0) return st;
  if (e != hipSuccess) return (int)e;
  *avg_ms = ms / (float)iters;
  return TFA_OK;
}

int tfa_set_variant(int variant) {
  // dispatchable numbers: -1 (automatic), the kVariants table, and the timing-only ablation ranges of the EXPERIMENTAL build
  const bool in_table = variant >= 0 && variant < tfa::kNumVariants;
  const bool ablation = (variant >= 100 && variant < 612) || (variant >= 700 && variant < 716) || (variant >= 1000 && variant < 2256) ||
                        (variant >= 3000 && variant < 3256);
  if (variant != -1 && !in_table && !ablation) return TFA_ERR_VARIANT;
  if (in_table && !tfa::variant_built(variant)) return TFA_ERR_VARIANT;
  g_variant = variant;
  return TFA_OK;
}
int tfa_get_variant(void) { return g_variant; }
int tfa_debug_set_flags(int flags) { g_dbg_flags = flags; return TFA_OK; }
int tfa_debug_decode(int B, int H, int Hk, int nwork, int id, int* out) {
  if (!out || B < 1 || H < 1 || Hk < 1 || nwork < 1 || id < 0 || H % Hk) return TFA_ERR_SHAPE;
  tfa::KArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.H = H; a.Hk = Hk; a.nwork = nwork; a.nbh = B * H;
  tfa::fill_decode(&a);
  // the device code of tfa_fwd_kernel_il.h, on the host (fd_div is a device function: its formula here)
  auto div = [](int n, tfa::FastDiv f) { return (int)(((unsigned)(((unsigned long long)(unsigned)n * f.m) >> 32) + (unsigned)n) >> f.l); };
  const int x = a.rr ? (id & 7) : 0, s = a.rr ? (id >> 3) : id;
  const int sq = div(s, a.fd_wa), r = s - sq * a.wa;
  const int kg = a.rr ? x + 8 * sq : sq;
  const int rq = div(r, a.fd_nwork);
  out[3] = r - rq * a.nwork;
  out[0] = div(kg, a.fd_wd);
  out[1] = (kg - out[0] * a.wd) * a.wg + rq;
  out[2] = div(out[1], a.fd_g);
  return TFA_OK;
}
int tfa_debug_set_trace(void* dev_buf) { g_trace = reinterpret_cast<unsigned long long*>(dev_buf); return TFA_OK; }
int tfa_num_variants(void) { return tfa::kNumVariants; }
int tfa_variant_available(int variant) { return tfa::variant_built(variant) ? 1 : 0; }
const char* tfa_variant_name(int variant) {
  if (variant < 0 || variant >= tfa::kNumVariants) return "auto";
  const tfa::Variant* v = tfa::variant_info(variant);
  return v ? v->name : "(an A/B arm: not in this build, make EXPERIMENTAL=1)";
}

int tfa_fwd_work(const tfa_fwd_params* p, double* flops, double* bytes) {
  if (!p) return TFA_ERR_NULL;
  const double bh = (double)p->B * p->H;
  double pairs;   // visible (query,key) pairs per (b,h)
  if (p->is_causal) {
    // row i sees min(Nk, max(0, i + 1 + Nk - Nq)) keys
    pairs = 0;
    const long long shift = (long long)p->Nk - p->Nq;
    if (shift >= 0) {
      pairs = (double)p->Nq * (double)(shift) + 0.5 * (double)p->Nq * ((double)p->Nq + 1.0);
    } else {
      const double n = (double)p->Nk;   // only the last Nk rows see anything
      pairs = 0.5 * n * (n + 1.0);
    }
    // the reference's convention is exactly half of the full square when Nq == Nk
    if (p->Nq == p->Nk) pairs = 0.5 * (double)p->Nq * (double)p->Nk;
  } else {
    pairs = (double)p->Nq * (double)p->Nk;
  }
  if (flops) *flops = 4.0 * bh * pairs * p->D;
  if (bytes) {
    const double osz = (p->out_dtype == TFA_F32) ? 4.0 : 2.0;
    const double bkv = (double)p->B * p->Hk;
    const double isz = (p->dtype == TFA_F32) ? 4.0 : 2.0;
    *bytes = bh * p->Nq * p->D * isz + 2.0 * bkv * p->Nk * p->D * isz + bh * p->Nq * p->D * osz +
             (p->lse ? bh * p->Nq * 4.0 : 0.0);
  }
  return TFA_OK;
}

}  // extern "C"
