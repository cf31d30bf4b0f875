// tfa_bwd_api.hip — extern "C" backward entry points of include/tfa.h: validate, fill kernel arguments, launch on the caller's stream
//   delta -> dQ (tfa_bwd_kernel.h) -> fused dK/dV (tfa_bwd_kv_kernel.h)                       the default: 7 GEMM units
//   delta -> fused dK/dV that also stores dS -> dQ = dS.K (tfa_bwd_dq_kernel.h)               with tfa_bwd_params::workspace: 5 units
//   delta -> dQ -> dK -> dV (tfa_bwd_kernel.h, three single-gradient launches)                head dims above 128, and the debug form
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "tfa.h"
#include "tfa_bwd_launch.h"

namespace tfa {
template <> hipError_t launch_bwd<__bf16, 64>(const BArgs&, int, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd<__bf16, 128>(const BArgs&, int, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd<_Float16, 64>(const BArgs&, int, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd<_Float16, 128>(const BArgs&, int, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd<__bf16, 256>(const BArgs&, int, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd<_Float16, 256>(const BArgs&, int, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_kv<__bf16, 64>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_kv<__bf16, 128>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_kv<_Float16, 64>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_kv<_Float16, 128>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_dq_ws<__bf16, 64>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_dq_ws<__bf16, 128>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_dq_ws<_Float16, 64>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_bwd_dq_ws<_Float16, 128>(const BArgs&, int, bool, bool, hipStream_t, bool);
template <> hipError_t launch_delta<__bf16, 64>(const void*, const void*, float*, const long long*, const long long*, int, int, long long, int, hipStream_t, bool);
template <> hipError_t launch_delta<__bf16, 128>(const void*, const void*, float*, const long long*, const long long*, int, int, long long, int, hipStream_t, bool);
template <> hipError_t launch_delta<_Float16, 64>(const void*, const void*, float*, const long long*, const long long*, int, int, long long, int, hipStream_t, bool);
template <> hipError_t launch_delta<_Float16, 128>(const void*, const void*, float*, const long long*, const long long*, int, int, long long, int, hipStream_t, bool);
template <> hipError_t launch_delta<__bf16, 256>(const void*, const void*, float*, const long long*, const long long*, int, int, long long, int, hipStream_t, bool);
template <> hipError_t launch_delta<_Float16, 256>(const void*, const void*, float*, const long long*, const long long*, int, int, long long, int, hipStream_t, bool);
}  // namespace tfa

namespace {

thread_local int g_bwd_split = 0;   // tfa_debug_bwd_split: bit 3 = delta by a launch of its own (not fused into the dQ launch), bit 0 = dK and dV as two launches (the round-1/2 form), bit 1 = force the windowed
                                    // (>= 2 GiB slices) instantiations on any problem; for A/B and parity cross-checks

// extent of one (b,h) slice.  Kernels with one descriptor per slice need every byte offset they form (up to 512 rows past the end)
// inside int32; the BIG instantiations (windowed descriptors, tfa_bwd_kernel.h) only need a window — 768 rows — to fit, and are
// launched when *big comes back set.  big == nullptr: the caller has no windowed form.
bool slice_bytes(int64_t n, int64_t row_stride, int d, int esize, unsigned* out, unsigned long long* full = nullptr, int* big = nullptr) {
  const int64_t bytes = ((n - 1) * row_stride + d) * esize;
  const int64_t reach = ((n + 512) * row_stride + d) * esize;   // every byte offset a kernel forms stays inside int32
  const int64_t window = (768 * row_stride + d) * esize;
  if (bytes <= 0) return false;
  if (reach >= (int64_t)0x7fffffff) {
    if (!big || window >= (int64_t)0x7fffffff) return false;
    *big = 1;
  }
  *out = (unsigned)(bytes < (int64_t)0x7fffffff ? bytes : (int64_t)0x7fffffff);
  if (full) *full = (unsigned long long)bytes;
  return true;
}

bool fill(tfa::BTensor* t, const void* ptr, const int64_t* st, int64_t n, int d, int esize, int* big) {
  t->p = ptr;
  t->s_b = st[0]; t->s_h = st[1]; t->s_n = st[2];
  return slice_bytes(n, st[2], d, esize, &t->bytes, &t->full, big);
}

int check_strides(const int64_t* st, int d, int esize) {
  for (int i = 0; i < 3; ++i) {
    if (st[i] < 0) return TFA_ERR_STRIDE;
    if ((st[i] * esize) % 16 != 0) return TFA_ERR_STRIDE;
  }
  if (st[2] < d) return TFA_ERR_STRIDE;
  return TFA_OK;
}

// bytes of the dS workspace for *p, 0 when a head's slab (roundup(Nk,128) x roundup(Nq,256) x 2 bytes) would not fit one descriptor
long long ws_bytes(const tfa_bwd_params* p, int* nk_pad, int* nq_pad) {
  const long long nk = ((long long)p->Nk + 127) / 128 * 128, nq = ((long long)p->Nq + 255) / 256 * 256;
  if (nk_pad) *nk_pad = (int)nk;
  if (nq_pad) *nq_pad = (int)nq;
  if (nk * nq * 2 >= (long long)0x7fffffff) return 0;
  return (long long)p->B * p->H * nk * nq * 2;
}

// ws_need (optional): receives the bytes of dS scratch the 5-GEMM form would use for *p, 0 when this problem never takes that form
int run_bwd(const tfa_bwd_params* p, void* stream, bool dry, long long* ws_need = nullptr) {
  if (ws_need) *ws_need = 0;
  if (!p) return TFA_ERR_NULL;
  if (!p->q || !p->k || !p->v || !p->out || !p->dout || !p->lse || !p->dq || !p->dk || !p->dv || !p->delta) return TFA_ERR_NULL;
  if (p->dtype != TFA_F16 && p->dtype != TFA_BF16) return TFA_ERR_DTYPE;
  if (p->grad_dtype != p->dtype && p->grad_dtype != TFA_F32) return TFA_ERR_DTYPE;
  if (p->D < 8 || p->D > 256 || (p->D % 8) != 0) return TFA_ERR_HEAD_DIM;   // kernels are 64, 128 and 256 wide; BArgs::dv = the valid part
  if (p->B <= 0 || p->H <= 0 || p->Hk <= 0 || p->Nq <= 0 || p->Nk <= 0 || p->H % p->Hk != 0) return TFA_ERR_SHAPE;
  if (!(p->softmax_scale > 0.f) || !isfinite(p->softmax_scale)) return TFA_ERR_SCALE;
  const int esz = 2, gsz = (p->grad_dtype == TFA_F32) ? 4 : 2;
  const int64_t* in_st[5] = {p->q_stride, p->k_stride, p->v_stride, p->o_stride, p->do_stride};
  for (int i = 0; i < 5; ++i) { const int st = check_strides(in_st[i], p->D, esz); if (st) return st; }
  const int64_t* g_st[3] = {p->dq_stride, p->dk_stride, p->dv_stride};
  for (int i = 0; i < 3; ++i) { const int st = check_strides(g_st[i], p->D, gsz); if (st) return st; }
  const uintptr_t al = (uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->out | (uintptr_t)p->dout | (uintptr_t)p->dq |
                       (uintptr_t)p->dk | (uintptr_t)p->dv;
  if (al & 15) return TFA_ERR_ALIGN;
  if (((uintptr_t)p->lse | (uintptr_t)p->delta) & 15) return TFA_ERR_ALIGN;   // read as 16-byte vectors
  if ((int64_t)p->B * p->H * p->Nq >= (int64_t)0x1fffffff) return TFA_ERR_SHAPE;

  tfa::BArgs a;
  memset(&a, 0, sizeof(a));
  // slices of 2 GiB and more (long (B,N,H,D) tensors): the windowed instantiations of the dQ launch and of the fused dK/dV launch —
  // not with the two-launch debug form at head dims up to 128 (its dK / dV launches have no windowed instantiation; the 256-wide kernel has one for each of its three launches)
  const bool can_big = p->D > 128 || !(g_bwd_split & 1);
  int big = (g_bwd_split & 2) ? 1 : 0;        // tests: the windowed instantiations on a small problem
  int* bigp = can_big ? &big : nullptr;
  if (!fill(&a.q, p->q, p->q_stride, p->Nq, p->D, esz, bigp)) return TFA_ERR_STRIDE;
  if (!fill(&a.k, p->k, p->k_stride, p->Nk, p->D, esz, bigp)) return TFA_ERR_STRIDE;
  if (!fill(&a.v, p->v, p->v_stride, p->Nk, p->D, esz, bigp)) return TFA_ERR_STRIDE;
  if (!fill(&a.dout, p->dout, p->do_stride, p->Nq, p->D, esz, bigp)) return TFA_ERR_STRIDE;
  {   // out (read by the delta kernel through plain 64-bit pointers) and the gradients: same rule
    unsigned tmp; unsigned long long tmpf;
    if (!slice_bytes(p->Nq, p->o_stride[2], p->D, esz, &tmp, &tmpf, bigp) || !slice_bytes(p->Nq, p->dq_stride[2], p->D, gsz, &tmp, &tmpf, bigp) ||
        !slice_bytes(p->Nk, p->dk_stride[2], p->D, gsz, &tmp, &tmpf, bigp) || !slice_bytes(p->Nk, p->dv_stride[2], p->D, gsz, &tmp, &tmpf, bigp))
      return TFA_ERR_STRIDE;
  }
  if (big && !can_big) return TFA_ERR_STRIDE;
  a.big = big;
  a.lse = p->lse; a.delta = p->delta;
  a.B = p->B; a.H = p->H; a.Hk = p->Hk; a.Nq = p->Nq; a.Nk = p->Nk;
  a.dv = p->D;
  const bool wide = p->D > 64;
  const bool wide256 = p->D > 128;          // head dims 136..256: one wave per SIMD, 128-row resident blocks, three single-gradient launches
  a.scale = p->softmax_scale;
  a.scale_log2 = p->softmax_scale * 1.4426950408889634f;
  const bool causal = p->is_causal != 0, f32 = p->grad_dtype == TFA_F32;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  auto launch = [&](int mode, void* grad, const int64_t* gst, int n_res, int h_res) -> int {
    tfa::BArgs m = a;
    m.grad = grad; m.gs_b = gst[0]; m.gs_h = gst[1]; m.gs_n = gst[2];
    if (!slice_bytes(n_res, gst[2], p->D, gsz, &m.g_bytes, &m.g_full, bigp)) return TFA_ERR_STRIDE;
    const int res_rows = wide256 ? 128 : 256;
    m.nrb = (n_res + res_rows - 1) / res_rows;
    const int64_t grid = (int64_t)p->B * h_res * m.nrb;
    if (grid >= (int64_t)0x7fffffff) return TFA_ERR_SHAPE;
    hipError_t e;
    if (p->dtype == TFA_BF16)
      e = wide256 ? tfa::launch_bwd<__bf16, 256>(m, mode, (int)grid, causal, f32, s, dry)
          : wide  ? tfa::launch_bwd<__bf16, 128>(m, mode, (int)grid, causal, f32, s, dry) : tfa::launch_bwd<__bf16, 64>(m, mode, (int)grid, causal, f32, s, dry);
    else
      e = wide256 ? tfa::launch_bwd<_Float16, 256>(m, mode, (int)grid, causal, f32, s, dry)
          : wide  ? tfa::launch_bwd<_Float16, 128>(m, mode, (int)grid, causal, f32, s, dry) : tfa::launch_bwd<_Float16, 64>(m, mode, (int)grid, causal, f32, s, dry);
    return (int)e;
  };

  // ---- with a workspace: dK/dV launch that also writes dS, then dQ = scale * dS . K (5 GEMM units) -----------------------------
  int nk_pad = 0, nq_pad = 0;
  const long long need = ws_bytes(p, &nk_pad, &nq_pad);
  if (p->workspace && (((uintptr_t)p->workspace) & 15)) return TFA_ERR_ALIGN;
#if defined(TFA_BWD_TRACE)
  // debug build: with bit 2 of tfa_debug_bwd_split the LAST 64 MiB of the workspace receive the fused launch's per-wave wait cycles
  constexpr long long kTraceBytes = 64ll << 20;
  const bool tracing = (g_bwd_split & 4) && p->workspace && p->workspace_bytes >= kTraceBytes;
  const long long ws_avail = p->workspace_bytes - (tracing ? kTraceBytes : 0);
  a.tr = tracing ? reinterpret_cast<char*>(p->workspace) + ws_avail : nullptr;
#else
  const long long ws_avail = p->workspace_bytes;
#endif
  // the kept-dS form exists for head dims up to 128, slices below 2 GiB, and not with the two-launch debug form
  const bool ws_form = need > 0 && !(g_bwd_split & 1) && !wide256 && !big;
  if (ws_need) *ws_need = ws_form ? need : 0;
  const bool use_ws = p->workspace != nullptr && ws_form && ws_avail >= need;
  // delta = rowsum(dout o out): the dQ launch computes it from its resident dO rows and the O rows and writes it for the launches behind it
  // (BArgs::fuse_delta; 51 us and one pass over dO less at config 3) — unless the fused dK/dV launch runs FIRST (the workspace form), or
  // tfa_debug_bwd_split value 8 (bit 3) asks for the launch of its own (A/B, tests)
  const bool fuse_delta = !use_ws && !(g_bwd_split & 8);
  // (always: the dQ launch reads O only when it forms delta, but fill() is also what validates out's strides and slice size for every form)
  if (!fill(&a.out, p->out, p->o_stride, p->Nq, p->D, esz, bigp)) return TFA_ERR_STRIDE;
  a.delta_w = p->delta;
  a.fuse_delta = fuse_delta ? 1 : 0;
  if (!fuse_delta)
  {
    const long long os[3] = {p->o_stride[0], p->o_stride[1], p->o_stride[2]};
    const long long ds[3] = {p->do_stride[0], p->do_stride[1], p->do_stride[2]};
    const long long rows = (long long)p->B * p->H * p->Nq;
    hipError_t e;
    if (p->dtype == TFA_BF16)
      e = wide256 ? tfa::launch_delta<__bf16, 256>(p->out, p->dout, p->delta, os, ds, p->H, p->Nq, rows, p->D, s, dry)
          : wide  ? tfa::launch_delta<__bf16, 128>(p->out, p->dout, p->delta, os, ds, p->H, p->Nq, rows, p->D, s, dry)
                  : tfa::launch_delta<__bf16, 64>(p->out, p->dout, p->delta, os, ds, p->H, p->Nq, rows, p->D, s, dry);
    else
      e = wide256 ? tfa::launch_delta<_Float16, 256>(p->out, p->dout, p->delta, os, ds, p->H, p->Nq, rows, p->D, s, dry)
          : wide  ? tfa::launch_delta<_Float16, 128>(p->out, p->dout, p->delta, os, ds, p->H, p->Nq, rows, p->D, s, dry)
                  : tfa::launch_delta<_Float16, 64>(p->out, p->dout, p->delta, os, ds, p->H, p->Nq, rows, p->D, s, dry);
    if (e != hipSuccess) return (int)e;
  }
  if (use_ws) {
    tfa::BArgs m = a;
    m.ws = p->workspace; m.ws_nk = nk_pad; m.ws_nq = nq_pad;
    m.grad = p->dk; m.gs_b = p->dk_stride[0]; m.gs_h = p->dk_stride[1]; m.gs_n = p->dk_stride[2];
    m.grad2 = p->dv; m.g2s_b = p->dv_stride[0]; m.g2s_h = p->dv_stride[1]; m.g2s_n = p->dv_stride[2];
    if (!slice_bytes(p->Nk, p->dk_stride[2], p->D, gsz, &m.g_bytes) || !slice_bytes(p->Nk, p->dv_stride[2], p->D, gsz, &m.g2_bytes)) return TFA_ERR_STRIDE;
    m.nrb = (p->Nk + 127) / 128;
    int64_t grid = (int64_t)p->B * p->Hk * m.nrb;
    if (grid >= (int64_t)0x7fffffff) return TFA_ERR_SHAPE;
    hipError_t e;
    if (p->dtype == TFA_BF16)
      e = wide ? tfa::launch_bwd_kv<__bf16, 128>(m, (int)grid, causal, f32, s, dry) : tfa::launch_bwd_kv<__bf16, 64>(m, (int)grid, causal, f32, s, dry);
    else
      e = wide ? tfa::launch_bwd_kv<_Float16, 128>(m, (int)grid, causal, f32, s, dry) : tfa::launch_bwd_kv<_Float16, 64>(m, (int)grid, causal, f32, s, dry);
    if (e != hipSuccess) return (int)e;
    tfa::BArgs d = a;
    d.ws = p->workspace; d.ws_nk = nk_pad; d.ws_nq = nq_pad;
    d.grad = p->dq; d.gs_b = p->dq_stride[0]; d.gs_h = p->dq_stride[1]; d.gs_n = p->dq_stride[2];
    if (!slice_bytes(p->Nq, p->dq_stride[2], p->D, gsz, &d.g_bytes)) return TFA_ERR_STRIDE;
    d.nrb = (p->Nq + 255) / 256;
    grid = (int64_t)p->B * p->H * d.nrb;
    if (grid >= (int64_t)0x7fffffff) return TFA_ERR_SHAPE;
    if (p->dtype == TFA_BF16)
      e = wide ? tfa::launch_bwd_dq_ws<__bf16, 128>(d, (int)grid, causal, f32, s, dry) : tfa::launch_bwd_dq_ws<__bf16, 64>(d, (int)grid, causal, f32, s, dry);
    else
      e = wide ? tfa::launch_bwd_dq_ws<_Float16, 128>(d, (int)grid, causal, f32, s, dry) : tfa::launch_bwd_dq_ws<_Float16, 64>(d, (int)grid, causal, f32, s, dry);
    return (int)e;
  }
  int st = launch(tfa::BWD_DQ, p->dq, p->dq_stride, p->Nq, p->H);
  if (st) return st;
  if ((g_bwd_split & 1) || wide256) {                            // head dims above 128, and debug / A-B: the two single-gradient launches (S computed twice)
    st = launch(tfa::BWD_DK, p->dk, p->dk_stride, p->Nk, p->Hk);
    if (st) return st;
    return launch(tfa::BWD_DV, p->dv, p->dv_stride, p->Nk, p->Hk);
  }
  // dK and dV in one launch: S and dP once each (tfa_bwd_kv_kernel.h)
  tfa::BArgs m = a;
  m.grad = p->dk; m.gs_b = p->dk_stride[0]; m.gs_h = p->dk_stride[1]; m.gs_n = p->dk_stride[2];
  m.grad2 = p->dv; m.g2s_b = p->dv_stride[0]; m.g2s_h = p->dv_stride[1]; m.g2s_n = p->dv_stride[2];
  if (!slice_bytes(p->Nk, p->dk_stride[2], p->D, gsz, &m.g_bytes, &m.g_full, bigp) ||
      !slice_bytes(p->Nk, p->dv_stride[2], p->D, gsz, &m.g2_bytes, &m.g2_full, bigp)) return TFA_ERR_STRIDE;
  constexpr int kv_keys = 32 * TFA_BWD_KV_KG_OF(false);   // resident keys per workgroup of the fused launch
  m.nrb = (p->Nk + kv_keys - 1) / kv_keys;
  const int64_t grid = (int64_t)p->B * p->Hk * m.nrb;
  if (grid >= (int64_t)0x7fffffff) return TFA_ERR_SHAPE;
  hipError_t e;
  if (p->dtype == TFA_BF16)
    e = wide ? tfa::launch_bwd_kv<__bf16, 128>(m, (int)grid, causal, f32, s, dry) : tfa::launch_bwd_kv<__bf16, 64>(m, (int)grid, causal, f32, s, dry);
  else
    e = wide ? tfa::launch_bwd_kv<_Float16, 128>(m, (int)grid, causal, f32, s, dry) : tfa::launch_bwd_kv<_Float16, 64>(m, (int)grid, causal, f32, s, dry);
  return (int)e;
}

}  // namespace

extern "C" {

int tfa_bwd(const tfa_bwd_params* p, void* stream) { return run_bwd(p, stream, false); }
int tfa_bwd_plan(const tfa_bwd_params* p) { return run_bwd(p, nullptr, true); }
int tfa_debug_bwd_split(int on) { g_bwd_split = on & 15; return TFA_OK; }
long long tfa_bwd_workspace_bytes(const tfa_bwd_params* p) {
  tfa_bwd_params q;
  if (!p) return TFA_ERR_NULL;
  q = *p;
  q.workspace = nullptr;
  long long need = 0;
  const int st = run_bwd(&q, nullptr, true, &need);     // 0 where run_bwd would ignore a workspace (D > 128, slices of 2 GiB and more, debug forms)
  if (st) return st;
  return need;
}

int tfa_bwd_work(const tfa_bwd_params* p, double* flops, double* bytes) {
  const int st = run_bwd(p, nullptr, true);
  if (st) return st;
  const double heads = (double)p->B * p->H, f = p->is_causal ? 0.5 : 1.0;
  if (flops) *flops = 10.0 * heads * p->Nq * (double)p->Nk * p->D * f;
  if (bytes) {
    const double gsz = p->grad_dtype == TFA_F32 ? 4.0 : 2.0;
    const double qrows = heads * p->Nq * p->D, krows = (double)p->B * p->Hk * p->Nk * p->D;
    *bytes = 2.0 * (3.0 * qrows + 2.0 * krows) + gsz * (qrows + 2.0 * krows) + 4.0 * heads * p->Nq;
  }
  return TFA_OK;
}

int tfa_bwd_time(const tfa_bwd_params* p, int warmup, int iters, void* stream, float* avg_ms) {
  if (!avg_ms || iters <= 0 || warmup < 0) return TFA_ERR_NULL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < warmup; ++i) { const int st = run_bwd(p, stream, false); if (st) return st; }
  hipEvent_t e0, e1;
  hipError_t e = hipEventCreate(&e0);
  if (e != hipSuccess) return (int)e;
  e = hipEventCreate(&e1);
  if (e != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
  int st = 0;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && st == 0; ++i) st = run_bwd(p, stream, false);
  (void)hipEventRecord(e1, s);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (st) return st;
  if (e != hipSuccess) return (int)e;
  *avg_ms = ms / (float)iters;
  return TFA_OK;
}

}  // extern "C"
