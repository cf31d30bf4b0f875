/*
 * tfa.h — C ABI of the MI355X (gfx950) FlashAttention-2 forward path.
 *
 * This is the drop-in boundary for the ONE hot path of 66RING/tiny-flash-attention:
 * the fused  S = scale * Q K^T  ->  online softmax  ->  O = P V  forward tile loop.
 * Everything here is `extern "C"`, plain pointers and sizes; no torch types.
 * Pointers are DEVICE pointers (HBM) unless stated otherwise.  The library never
 * allocates or frees device memory and keeps no state between calls except the
 * PER-THREAD debug knobs tfa_set_variant() / tfa_debug_set_trace() (thread-local: a thread
 * that forces a variant does not change what other threads' calls run).
 *
 * Reference interfaces each entry point replaces (paths relative to the reference repo):
 *
 *   tfa_fwd_bhnd      <- flash_attention_v2_cutlass(q,k,v,is_causal,softmax_scale) -> {out, lse}
 *                        flash_attention_cutlass/csrc/flash_attention.cu:741-772
 *                        (declared flash_attention_cutlass/include/attention_api.h:10-11,
 *                         bound   flash_attention_cutlass/csrc/attention_api.cpp:6-10)
 *                     <- flash_attention_v2_cuda(q,k,v) -> out      [scale=1/sqrt(D), non causal]
 *                        flash_attention_cuda/csrc/flash_attention.cu:375-424
 *                     <- _kernels.flash_attn(q,k,v,is_causal,softmax_scale) -> out   [CPU sibling]
 *                        flash_attention_c/csrc/attn.cpp:237-262
 *   tfa_fwd           <- set_params_fprop + run_flash_attn_cutlass (strided / GQA / Nq!=Nk general form)
 *                        flash_attention_cutlass/csrc/flash_attention.cu:320-361, 731-739
 *                        flash_attention_cutlass/csrc/flash.h:6-59   (Flash_fwd_params)
 *                        flash_attention_c/csrc/attn.cpp:171-203     (strided params, causal offset)
 *   tfa_strerror      <- CUDA_ERROR_CHECK / TORCH_CHECK text
 *                        flash_attention_cutlass/include/attention_api.cuh:12-29
 *   tfa_merge, tfa_fwd_splitkv (partial results over key chunks + merge)
 *                     <- the v1 block-merge rule  flash_attention_py/tiny_flash_attn.py:63-68, README_zh.md:104-125
 *   tfa_bwd           <- no reference entry: the reference only SAVES softmax_lse for a backward
 *                        flash_attention_cutlass/csrc/flash_attention.cu:353-354, 614-623
 *
 * Semantics (identical to the reference; see oracle/ for the CPU restatement):
 *   S[i,j]  = softmax_scale * sum_d q[i,d] k[j,d]            (16-bit products, fp32 accumulate)
 *   causal:   S[i,j] = -inf for j > i + (Nk - Nq)            (attn.cpp:122-124)
 *   m_i = max_j S ; P = exp(S - m_i) ; l_i = sum_j P         (fp32)
 *   O[i,:]  = (sum_j round16(P[i,j]) v[j,:]) / l_i           (P rounded to the input dtype before PV,
 *                                                             flash_attention.cu:601; l sums unrounded P)
 *   O -> rounded to the input dtype (RNE), or left fp32 when out_dtype == TFA_F32
 *   LSE_i   = m_i + ln(l_i)   (natural log, scale included)  (flash_attention.cu:623)
 *   empty row (no visible key): O = 0, LSE = +inf            (flash_attention.cu:620-623)
 * Rounding points (which row reference m' stands in P = exp(S - m') when P is rounded to 16 bits; O and LSE are the same numbers mathematically
 * under every rule, and every rule holds the reference's atol 1e-2 and the rigorous bound |O - O_fp64| <= 2^-8 * A (bf16) / 2^-11 * A (fp16),
 * A[i,d] = sum_j P[i,j] |v[j,d]| / l_i — tfa_fwd_rounding_rule says which one a call runs):
 *   TFA_RULE_EXACT_MAX   the exact running maximum, the reference's own rule (flash_attention.cu:263-316, main_torch_only.py:240-260): the flag
 *                        TFA_FWD_EXACT_MAX (element-wise rtol 1e-3 against the reference's tile loop), the split-KV kernel, fp32 tensors
 *   TFA_RULE_LAZY        a per-row reference that trails the running maximum by at most 2^8 (P <= 2^8; a wave of 32 rows re-bases together when
 *                        one of them outgrows it): fp16 on the default kernels, every special-case instantiation, head dims above 128
 *   TFA_RULE_FIRST_TILE  bf16 on the default kernels' main instantiations (round 6): m' = the row's maximum over its FIRST 64-key tile, kept for
 *                        the whole row — bf16 has fp32's exponent range, so no running maximum is needed for range, and the kernel forms none
 *                        (it tests the row SUMS instead: beyond 2^40 a wave re-bases by an exact power of two, which moves no result bit; a
 *                        single tile that lifts a row sum beyond 2^64 makes the workgroup redo that query block with TFA_RULE_LAZY).
 *                        Stated domain: |v| * Nk < 2^63 (beyond it P*v could overflow fp32 before a re-base; the lazy rule's own limit is 2^119).
 *
 * Error convention: every entry point returns 0 on success, a negative tfa_status on a
 * rejected argument (nothing was launched), or a positive hipError_t when the HIP runtime
 * reported an error on launch.  No entry point synchronises the device or calls exit()
 * (the reference does both, flash_attention.cu:767-769; callers already synchronise).
 */
#ifndef TFA_H_
#define TFA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFA_VERSION 111 /* 0.1.11: tfa_bwd at head dims up to 128 runs hand-scheduled tile loops in both launches; the dQ launch accumulates dP from -delta (gradient bits differ from 0.1.10, inside the same bounds); no interface change.  0.1.10: bf16 on the default kernels rounds P against the first key tile's row maximum (max-free tile loop; tfa_fwd_rounding_rule says which rule a call runs), the tile bodies behind the hand-scheduled loop are generated too, tfa_debug_mfma_ceiling returns TFA_ERR_SHAPE for bad sizes; 0.1.9: (b,h) slices of 2 GiB and more at head dims above 128 (windowed instantiations of the 256-wide forward and backward kernels; TFA_ERR_STRIDE before), tfa_debug_mfma_ceiling; 0.1.8: TFA_FWD_EXACT_MAX runs the il8 kernel's exact-max instantiation (variant 38) on grids that fill the chip; 0.1.7: tfa_bwd computes delta inside its dQ launch (tfa_debug_bwd_split bit 3 restores the separate launch), tfa_debug_set_trace is served by traced twins of the main kernels; 0.1.6: + fp32 q,k,v (TFA_F32 input: the correctness path behind the reference's fp32 fixtures), variant numbers are ids (tfa_variant_available), tfa_bwd_workspace_bytes is 0 wherever the workspace would be ignored; 0.1.5: + tfa_fwd_params::flags (TFA_FWD_EXACT_MAX), split-KV for head dims up to 256; 0.1.4: + tfa_fwd_suggest_splits, key-split kernels for small grids, split-KV / backward head dims multiples of 8; 0.1.3: forward head dims = every multiple of 8 up to 256; tfa_debug_set_flags; 0.1.2: + tfa_variant_available; debug knobs are per thread; 0.1.1: split-KV, tfa_merge, tfa_bwd */

/* element types */
enum tfa_dtype { TFA_F16 = 0, TFA_BF16 = 1,
                 TFA_F32 = 2 /* outputs of the 16-bit kernels; as the dtype of q,k,v: the fp32 correctness path (below) */ };

enum tfa_status {
  TFA_OK = 0,
  TFA_ERR_NULL = -1,          /* a required pointer is NULL */
  TFA_ERR_DTYPE = -2,         /* dtype not in {F16,BF16}; out_dtype not in {dtype,F32} */
  TFA_ERR_HEAD_DIM = -3,      /* forward, split-KV, backward: D not a multiple of 8 in [8,256]; merge: not a multiple of 4 in [4,256];
                               * TFA_FWD_EXACT_MAX: not a multiple of 8 in [8,128] */
  TFA_ERR_SHAPE = -4,         /* B,H,Hk,Nq,Nk <= 0 or H % Hk != 0 */
  TFA_ERR_STRIDE = -5,        /* a stride is negative, not 16-byte aligned, rows overlap, or 768 rows of a (b,h) slice span 2 GiB
                               * (tfa_fwd and tfa_bwd switch to per-block / per-tile descriptor windows when a slice is larger, ~6 % / ~3 %
                               * slower — at every head dim since 0.1.9; tfa_fwd_splitkv then runs one windowed launch per key chunk) */
  TFA_ERR_ALIGN = -6,         /* a base pointer is not 16-byte aligned */
  TFA_ERR_VARIANT = -7,       /* unknown kernel variant */
  TFA_ERR_SCALE = -8          /* softmax_scale is not finite or is <= 0 */
};

/*
 * General problem descriptor.  Tensors are 4-D logical (B, H, N, D) with arbitrary
 * batch/head/row strides (in ELEMENTS) and unit stride along D, so both the reference's
 * contiguous (B,H,N,D) layout and the (B,N,H,D) layout of flash_attn_func are expressible.
 * K and V have Hk heads (Hk divides H; Hk == H for plain MHA; query head h reads kv head
 * h / (H/Hk), the grouping of flash_attention_c/csrc/archive_)/attn.cpp:61).
 */
typedef struct tfa_fwd_params {
  const void* q;  /* (B,H ,Nq,D) */
  const void* k;  /* (B,Hk,Nk,D) */
  const void* v;  /* (B,Hk,Nk,D) */
  void* out;      /* (B,H ,Nq,D) of out_dtype */
  float* lse;     /* (B,H,Nq) fp32 contiguous, or NULL to skip */
  int32_t B, H, Hk, Nq, Nk, D;
  int64_t q_stride[3];   /* batch, head, row (elements) */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int64_t o_stride[3];   /* in elements of out_dtype */
  float softmax_scale;
  int32_t is_causal;     /* bottom-right aligned when Nq != Nk */
  int32_t dtype;         /* tfa_dtype of q,k,v: TFA_F16 or TFA_BF16 (the MFMA kernels: everything this header describes), or TFA_F32 — fp32
                          * tensors, the dtype of the reference's own CPU fixtures (flash_attention_c/test.py:35-48) and of the float arm of
                          * flash_attention_cuda/csrc/flash_attention.cu:411: served by a correctness kernel with fp32 arithmetic end to end
                          * (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate; csrc/tfa_fwd_f32.hip), out_dtype must be TFA_F32, head dims =
                          * multiples of 4 up to 256, rows 16-byte aligned, flags 0; tfa_fwd only (no split-KV, no backward).  Meets the
                          * reference's fp32 results to 1e-5 (tests/test_f32_gpu.py). */
  int32_t out_dtype;     /* == dtype, or TFA_F32 (debug/parity: unrounded fp32 O; also the partial results of split-KV) */
  /* split-KV: when k, v are the chunk [kv_offset, kv_offset + Nk) of a longer key
   * sequence of nk_total keys, the causal mask is taken against GLOBAL key positions: key kv_offset + j is
   * visible to row i iff kv_offset + j <= i + (nk_total - Nq).  out / lse are then PARTIAL results (rows that
   * see no key of the chunk: out = 0, lse = +inf) to be combined with tfa_merge.  Both 0 = the whole sequence. */
  int64_t kv_offset;
  int64_t nk_total;      /* 0 means kv_offset + Nk */
  int32_t flags;         /* TFA_FWD_* bits, 0 = default */
  int32_t reserved_;     /* must be 0 */
} tfa_fwd_params;

/* tfa_fwd_params::flags
 * TFA_FWD_EXACT_MAX: round P to 16 bits at the REFERENCE's points — every KV tile is exponentiated against the exact running
 *   row maximum, as flash_attention_cutlass/csrc/flash_attention.cu:263-316 and flash_attention_py/main_torch_only.py:240-260
 *   do — instead of the default kernels' own row reference (TFA_RULE_LAZY / TFA_RULE_FIRST_TILE above: same mathematics, O and LSE agree to
 *   the P-rounding bound, but the 16-bit roundings of P fall elsewhere).  Head dims up to 128, (b,h) slices below 2 GiB, no GQA row packing.  Where the
 *   default would run the il8 kernel (grids that fill the chip: the BASELINE configs 3, 4, 5) the flag runs that kernel's exact-max
 *   instantiation ("exact-il8", variant 38, round 5: the same issue-interleaved tile body; a tile in which some row of a wave saw a new
 *   maximum also multiplies O by exp2(old - new) behind its QK^T MFMAs) — 8-10 % slower than the default: bench.py quotes the pair in one line,
 *   `value` and `value_at_reference_rounding_points`; everywhere else the burst-structured LDS-DMA kernel (variant 17; 10-15 % slower than the default).  For callers
 *   that compare against the reference element by element (rtol 1e-3 with fp32 output).
 *
 * WHICH TOLERANCE EACH PATH GUARANTEES (stated and asserted in tests/test_parity_gpu.py; A[i,d] = sum_j P_ij |v_jd| is the
 * non-cancelling magnitude of an output element, eps16 = 2^-8 for bf16, 2^-11 for fp16):
 *   every path, 16-bit output vs the exact (fp64) result:      |d| <= 1e-2                 — the reference's own bar (test.py:87)
 *   every path, fp32 output vs the exact result:               |d| <= eps16 * A + 1e-6     — the rigorous bound of rounding P to 16 bits
 *   every path, LSE:                                           |d| <= 1e-4, +inf exactly where a row sees no key
 *   default kernels (TFA_RULE_LAZY / TFA_RULE_FIRST_TILE), fp32 output vs the reference's tile loop restated with THEIR rounding points:
 *                                                              |d| <= 1e-3 * |ref| + 1e-4 * A  (<= 1e-4 of the elements may flip one rounding of P)
 *   TFA_FWD_EXACT_MAX, fp32 output vs the reference's own tile loop (main_torch_only.py:160-270), element by element:
 *                                                              |d| <= 1e-3 * |ref|  on the elements with |ref| > 0.05 * A (config 4: 0.01 * A) — an element
 *                                                              whose value is a cancelling sum has no meaningful relative error —, 1e-4 of those
 *                                                              elements exempt: this is how BASELINE.json's rtol = 1e-3 is read and asserted
 *                                                              (tests/test_parity_gpu.py::test_exact_running_max_flag_at_baseline_sizes, whole heads of
 *                                                              BASELINE configs 3 and 4); measured cost: 0.517-0.529 vs 0.468-0.479 ms on the headline
 *                                                              shape (profiles/r06_bench_driver_protocol*.json). */
#define TFA_FWD_EXACT_MAX 1

/* Library version (TFA_VERSION of the build). */
int tfa_version(void);

/* Human-readable text for a return code of any entry point (static storage). */
const char* tfa_strerror(int status);

/* Launch the forward pass described by *p on HIP stream `stream` (NULL = default stream).
 * Asynchronous: returns after enqueueing.  Never allocates.  The kernel is chosen from the problem's size (256-row blocks,
 * 128-row blocks, keys split inside the workgroup for grids smaller than the chip, the 256-wide kernel for D > 128); with
 * Hk < H and one query row per head (batched decode) the H/Hk query heads of a K/V head are run as rows of one problem, so
 * K and V stream once per K/V head.  Results do not depend on those choices beyond the rounding of P to 16 bits. */
int tfa_fwd(const tfa_fwd_params* p, void* stream);

/* Convenience form for the reference's layout: q,k,v,out contiguous (B,H,N,D), Nq == Nk == N,
 * Hk == H, out dtype == input dtype.  This is exactly the argument list of
 * flash_attention_v2_cutlass (flash_attention.cu:741-742) plus explicit sizes. */
int tfa_fwd_bhnd(const void* q, const void* k, const void* v, void* out, float* lse,
                 int B, int H, int N, int D, float softmax_scale, int is_causal,
                 int dtype, void* stream);

/* Same, but O is written as fp32 (no final rounding) — the debug path used to check the
 * rtol=1e-3 parity target below one bf16 ulp (precedent: FPC_O=float,
 * flash_attention_cutlass/standalone_src/flash_attention_cutlass_standalone.cu:18-23). */
int tfa_fwd_bhnd_f32out(const void* q, const void* k, const void* v, float* out, float* lse,
                        int B, int H, int N, int D, float softmax_scale, int is_causal,
                        int dtype, void* stream);

/* Validate *p without launching; on success optionally reports the launch geometry. */
int tfa_fwd_plan(const tfa_fwd_params* p, int* grid, int* block, int* lds_bytes);

/* The kernel variant tfa_fwd would run for *p (>= 0; the forced one if tfa_set_variant is active), or a
 * negative TFA_ERR_* code.  The parity tests use it to pick the matching same-rounding-points emulation:
 * variants named "il..." keep a lazily re-based row reference instead of the exact running max
 * (oracle/oracle.py: tiled_emulation_lazy), all others follow main_torch_only.py:240-257 exactly. */
int tfa_fwd_variant(const tfa_fwd_params* p);

/* The row reference tfa_fwd would round P against for *p (the header's "Rounding points"): TFA_RULE_*, or a negative TFA_ERR_* code.  Decided
 * by the same predicates as the launch (kernel variant, dtype, which instantiation of the variant the sizes select); the parity tests pick
 * their same-rounding-points emulation with it (oracle/oracle.py: tiled_emulation, tiled_emulation_lazy, tiled_emulation_first_tile). */
#define TFA_RULE_EXACT_MAX 0
#define TFA_RULE_LAZY 1
#define TFA_RULE_FIRST_TILE 2
int tfa_fwd_rounding_rule(const tfa_fwd_params* p);

/* Time `iters` back-to-back launches of *p with HIP events recorded on `stream`
 * (after `warmup` untimed launches).  Writes the average milliseconds per launch.
 * Synchronises `stream`.  Used by bench.py for the roofline numbers. */
int tfa_fwd_time(const tfa_fwd_params* p, int warmup, int iters, void* stream, float* avg_ms);

/* ---- split-KV merge ---------------------------------------------------------
 * Combines `nparts` partial attention results over disjoint key chunks of the same queries — the online-softmax
 * merge rule the reference states for its v1 form (flash_attention_py/tiny_flash_attn.py:63-68, README_zh.md:104-125):
 *   lse = log sum_p exp(lse_p),   out = sum_p exp(lse_p - lse) * out_p      (parts with lse_p = +inf are empty).
 * o_parts: nparts x rows x D fp32 (part stride o_part_stride elements, rows contiguous, rows = B*H*Nq);
 * lse_parts: nparts x rows fp32 (part stride lse_part_stride).  out: rows x D of out_dtype (F16/BF16/F32),
 * lse_out: rows fp32 or NULL.  A row whose parts are all empty gives out = 0, lse = +inf. */
int tfa_merge(const float* o_parts, const float* lse_parts, int nparts, int64_t rows, int D,
              int64_t o_part_stride, int64_t lse_part_stride, void* out, int out_dtype, float* lse_out, void* stream);

/* Split-KV on one GPU in ONE launch (decode-like shapes: few query rows, long K/V, too few workgroups to fill the
 * chip): the key sequence is cut into `splits` chunks (multiples of 64 keys), the grid carries one copy of the work per
 * chunk (LDS-DMA kernel, 64 / 128 / 256 wide), partials go to `workspace`, tfa_merge writes *p's out (contiguous (B,H,Nq,D)) and lse.
 * (b,h) slices of 2 GiB and more: one launch of tfa_fwd's windowed kernel per chunk instead; when such a launch leaves
 * the chip mostly idle the launches are forked over four side streams owned by the calling thread and joined into `stream` before the
 * merge (event fork / join: legal inside a stream capture — the streams and events are created on the thread's first such call, so make
 * one call outside a capture first).  Everything the call enqueues is ordered before later work on `stream`.
 * workspace: tfa_fwd_splitkv_workspace(p, splits) floats (16-byte aligned); negative return = TFA_ERR_*. */
long long tfa_fwd_splitkv_workspace(const tfa_fwd_params* p, int splits);
int tfa_fwd_splitkv(const tfa_fwd_params* p, int splits, float* workspace, void* stream);
/* The split count a host that can provide a workspace should use for *p: 1 = call tfa_fwd (the grid fills the chip, or the
 * keys are too few to be worth a merge), >= 2 = call tfa_fwd_splitkv with that many chunks (decode-like shapes: B*H*ceil(Nq/128)
 * workgroups on half of the CUs or fewer, at least 4096 keys, causal only when Nq <= Nk/4; measured 3-9x on B1 H32 Nq1
 * Nk16k..64k, B1 H8 Nq16 Nk32k, 1.7-2x on short non-causal / chunked-prefill problems with one to four heads, 1.4-1.6x when a quarter to a half of the CUs had work; head dims above 128:
 * 4-11x, K/V at 3.6-5.5 TB/s; at most 4 for the one-launch-per-chunk route of slices beyond 2 GiB: 1.4-1.8x).
 * The reference-named bindings (attention_cutlass / attention_cuda / _kernels and their Python mirrors) follow it. */
int tfa_fwd_suggest_splits(const tfa_fwd_params* p);

/* ---- backward ------------------------------------------------------------
 * The reference has no backward pass; it saves softmax_lse for one ("LogSumExp save for backward",
 * flash_attention_cutlass/csrc/flash_attention.cu:353-354, :614-623; tiny_flash_attn_triton.py:27-29).
 * tfa_bwd consumes exactly what tfa_fwd produced: out and lse of the same q,k,v, and the upstream
 * gradient dout (same shape/dtype as out), and writes dq (shape of q), dk, dv (shape of k, v; for
 * grouped-query attention summed over the query heads of each kv head).  Same layout rules as tfa_fwd
 * (strides in elements, unit stride along D, 16-byte aligned rows); grads are written in the input
 * dtype, or in fp32 when grad_dtype == TFA_F32 (debug path for tolerance checks below one 16-bit ulp).
 * `delta` is caller-provided scratch of B*H*Nq floats (rowsum(dout o out), filled by the call). */
typedef struct tfa_bwd_params {
  const void* q;
  const void* k;
  const void* v;
  const void* out;
  const void* dout;
  const float* lse;      /* (B,H,Nq) contiguous, natural log, as written by tfa_fwd */
  void* dq;
  void* dk;
  void* dv;
  float* delta;          /* scratch, B*H*Nq floats */
  int B, H, Hk, Nq, Nk, D;
  int64_t q_stride[3];   /* batch, head, row (elements) */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int64_t o_stride[3];
  int64_t do_stride[3];
  int64_t dq_stride[3];
  int64_t dk_stride[3];
  int64_t dv_stride[3];
  float softmax_scale;
  int is_causal;
  int dtype;             /* TFA_F16 / TFA_BF16: q,k,v,out,dout */
  int grad_dtype;        /* == dtype, or TFA_F32 */
  /* Optional: a scratch buffer of at least tfa_bwd_workspace_bytes(p) bytes (16-byte aligned), or NULL.  With it the dK/dV launch
   * keeps dS = P o (dP - delta) (16 bit, B*H*Nk*Nq elements) and dQ becomes ONE GEMM over it instead of a recomputation of S and
   * dP: 5 GEMM units in all instead of 7, at the price of O(Nq*Nk) scratch memory.  Same results (the same 16-bit dS feeds dQ
   * either way, up to P having been rounded to 16 bit first), still deterministic.  NULL / too small: the O(N)-memory path. */
  void* workspace;
  int64_t workspace_bytes;
} tfa_bwd_params;

/* Launch the backward on `stream` (asynchronous): dQ (S, dP, dQ: 3 GEMM units; the launch also computes delta = rowsum(dO o O) for its rows
 * and writes it to tfa_bwd_params::delta), then dK and dV in ONE launch that computes S and dP once each (4 units; tfa_bwd_kv_kernel.h);
 * with tfa_bwd_params::workspace: delta (a launch of its own), dK/dV (which also writes dS), dQ = dS.K (1 unit).
 * Head dims 136..256: three single-gradient launches (dQ, dK, dV) of the 256-wide kernel, one wave per SIMD.
 * Deterministic: no atomics, fixed summation order. */
int tfa_bwd(const tfa_bwd_params* p, void* stream);
/* Debug / A-B (per thread): bit 0 makes tfa_bwd run dK and dV as two single-gradient launches (S computed twice: the form of
 * versions <= 0.1.4); bit 1 forces the windowed instantiations (the ones slices of 2 GiB and more get) on any problem; bit 3 (value 8)
 * computes delta by a launch of its own in front of the dQ launch (the form up to version 0.1.6) instead of inside it. */
int tfa_debug_bwd_split(int on);
/* Validate *p without launching (no GPU needed). */
int tfa_bwd_plan(const tfa_bwd_params* p);
/* Bytes of tfa_bwd_params::workspace that switch tfa_bwd to its 5-GEMM form for *p (B*H * roundup(Nk,128) * roundup(Nq,256) * 2),
 * 0 when tfa_bwd would not use a workspace for *p (head dims above 128, (b,h) slices of 2 GiB and more, a head's slab reaching
 * 2 GiB, the two-launch debug form), negative = TFA_ERR_*. */
long long tfa_bwd_workspace_bytes(const tfa_bwd_params* p);
/* Algorithmic work of one call: flops = 2.5 x the forward's (5 GEMMs of 2*Nq*Nk*D each per head, halved
 * when causal), bytes = q,k,v,out,dout read once + dq,dk,dv written once + lse. */
int tfa_bwd_work(const tfa_bwd_params* p, double* flops, double* bytes);
/* Time `iters` back-to-back tfa_bwd calls with HIP events on `stream` (after `warmup` untimed ones). */
int tfa_bwd_time(const tfa_bwd_params* p, int warmup, int iters, void* stream, float* avg_ms);

/* Kernel-variant selector for A/B measurement and bring-up (state of the CALLING THREAD).  -1 = automatic (default).
 * Variant numbers live in [0, tfa_num_variants()); tfa_variant_available(i) says whether THIS build carries number i (the product
 * build: the six dispatched kernels 17, 30, 32, 34, 36, 37), tfa_variant_name(i) describes it. */
int tfa_set_variant(int variant);
int tfa_get_variant(void);
int tfa_num_variants(void);
const char* tfa_variant_name(int variant);
/* 1 if variant i is compiled into this build, else 0.  The product build carries the dispatched kernels only; the
 * other entries of the table are A/B arms built with -DTFA_EXPERIMENTAL (make EXPERIMENTAL=1) and are rejected
 * with TFA_ERR_VARIANT otherwise. */
int tfa_variant_available(int variant);

/* Debug/profiling: when dev_buf != NULL every workgroup of subsequent launches writes 8 x uint64
 * {t_start, t_after_prologue, t_after_loop, t_end (shader cycles, s_memtime), n_kv_tiles (low 32 bits),
 * XCC_ID | HW_ID << 32, the workgroup's life in 100 MHz s_memrealtime ticks, (bh<<32)|query_block}
 * at dev_buf[8*workgroup_id ...]; the buffer must hold 64 B per workgroup (tfa_fwd_plan reports the
 * grid).  (t_end - t_start) / ticks * 100 MHz is the shader clock the workgroup actually ran at:
 * bench.py reports its median as the sustained clock next to the nominal 2.4 GHz.  NULL (default)
 * disables it.  The stamps are written by a traced TWIN of the kernel (the kernels the library dispatches are compiled without the
 * stamp code: its scalar state costs every workgroup ~130 register-lane moves): the main 16-bit-output instantiation of the il kernels
 * (variants 30, 32, 36, 37), and the dma / x4 kernels; the special-case instantiations (windowed slices, decode-like idle waves, head
 * dims below the kernel's width, fp32 output) leave the buffer untouched. */
int tfa_debug_set_trace(void* dev_buf);
/* Kernel bring-up flags of the calling thread (0 = normal).  128: the trace stamps describe a causal workgroup's SECOND
 * pass (the light block) instead of the first; 256: launch the windowed-descriptor instantiation (the one slices of
 * 2 GiB and more get) whatever the slice size — tests compare its bits with the default; 8192: tfa_fwd_splitkv takes its one-launch-per-chunk
 * route (the one (b,h) slices of 2 GiB and more take) on any problem; 16384: that route launches its chunks in
 * line on the caller's stream instead of forking them over the thread's side streams; the low bits insert fences / force the burst path in the x4 kernel
 * (tfa_fwd_kernel_x4.h) and are only meaningful to tools/. */
int tfa_debug_set_flags(int flags);
/* Host-side check of the kernels' work-item decode (no GPU needed): decodes workgroup `id` of a launch with the given batch, query
 * heads, K/V heads and work items per head exactly as the il kernels do on the device — the host-computed magic-number divisions
 * (tfa_launch.h: fill_decode, tfa_fwd_kernel.h: FastDiv) applied in the kernels' branch-free form — and writes {b, h, hk, wi} to
 * out[0..3].  tests/test_abi.py compares it with the plain divisions of the three dispatch orders. */
int tfa_debug_decode(int B, int H, int Hk, int nwork, int id, int* out);

/* Measurement aid (bench.py): the rate in TFLOP/s that a stream of nothing but v_mfma_f32_32x32x16_bf16 sustains on this GPU with operand values
 * taken from the first 16 MiB of `operands` (device memory, `bytes` >= 16 MiB of bf16 data — bench.py passes its q tensor), launched on `stream` for about `seconds`
 * (<= 30); synchronises the stream.  On the reference's normal(0, 0.5) inputs the board's power cap holds that stream to ~0.68 of the nominal
 * 2.5 PFLOP/s and boxes differ by +-5 %: the figure belongs next to `roofline.frac`, measured in the same run, not hard-coded.
 * It SYNCHRONISES `stream` between its groups of launches and allocates / frees device memory: it cannot be called inside a stream capture.
 * *tflops is 0 on every error return (bad sizes: TFA_ERR_SHAPE; no timed group completed: hipErrorNotReady). */
int tfa_debug_mfma_ceiling(const void* operands, unsigned long long bytes, double seconds, void* stream, double* tflops);

/* Algorithmic work of *p: flops = 4*B*H*Nq*Nk*D (x1/2 when causal, the reference's
 * convention) and bytes = Q+K+V read once + O written once (+LSE). */
int tfa_fwd_work(const tfa_fwd_params* p, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* TFA_H_ */
