/*
 * oracle/ref_attn.c — CPU restatement of the reference's attention forward.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing on the product path (tiny-flash-attention_amd/) may
 * import, link or call this file; it is the checker for tests/, __graft_entry__.smoke()
 * and the `cpu_baseline` leg of bench.py.
 *
 * Each function cites the reference lines it follows (paths relative to the reference repo).
 * Parity of this restatement is pinned by tests/test_oracle.py against
 *   (a) oracle/_ref/_kernels*.so — the reference's own flash_attention_c sources compiled
 *       unmodified from /root/reference (oracle/Makefile), when present, and
 *   (b) tests/golden/*.npz — outputs of the reference's Python/C implementations on the
 *       reference's own fixture recipes (tests/golden/make_golden.py).
 *
 * Build:  make -C oracle        (gcc -O3 -fopenmp, plain C99)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* round-to-nearest-even fp32 -> bf16 -> fp32 */
static float round_bf16(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&x, &u, 4);
  return x;
}

/* round-to-nearest-even fp32 -> fp16 -> fp32 (values in [0,1] only need the normal/subnormal path) */
static float round_fp16(float x) {
#if defined(__FLT16_MANT_DIG__)
  return (float)(_Float16)x;
#else
  /* portable fallback: scale so the fp32 rounding happens at the fp16 ulp */
  if (x == 0.f || !isfinite(x)) return x;
  int e;
  frexpf(x, &e);               /* x = f * 2^e, 0.5 <= |f| < 1 */
  int shift = 11 - e;          /* 11 significant bits */
  if (e < -13) shift = 24;     /* subnormal fp16: fixed ulp 2^-24 */
  float s = ldexpf(1.f, shift);
  return nearbyintf(x * s) / s;
#endif
}

static float round_p(float x, int mode) {
  if (mode == 1) return round_bf16(x);
  if (mode == 2) return round_fp16(x);
  return x;
}

typedef struct {
  const float* q; const float* k; const float* v; float* o; float* lse;
  int B, H, Hk, Nq, Nk, D;
  int64_t qs[3], ks[3], vs[3], os[3];   /* batch, head, row strides in elements */
  float scale; int causal;
} oracle_args;

/* number of keys row i may see: reference causal offset, flash_attention_c/csrc/attn.cpp:50-53,121-124 */
static int kv_len_of(const oracle_args* a, int i) {
  int kv_len = a->Nk;
  if (a->causal) kv_len = i + 1 + (a->Nk - a->Nq);
  if (kv_len < 0) kv_len = 0;
  if (kv_len > a->Nk) kv_len = a->Nk;
  return kv_len;
}

/*
 * Two-pass ("naive") attention in fp32: follows run_naive_attn,
 * flash_attention_c/csrc/attn.cpp:35-98 — scores, row max, exp, normalise, then P@V.
 * (The reference normalises P before the PV product; so does this.)
 */
int oracle_attn_naive(const oracle_args* a) {
  const int D = a->D, g = a->H / a->Hk;
#pragma omp parallel
  {
    float* score = (float*)malloc(sizeof(float) * (size_t)(a->Nk > 0 ? a->Nk : 1));
#pragma omp for collapse(3) schedule(dynamic, 8)
    for (int b = 0; b < a->B; b++)
      for (int h = 0; h < a->H; h++)
        for (int i = 0; i < a->Nq; i++) {
          const float* q = a->q + b * a->qs[0] + h * a->qs[1] + i * a->qs[2];
          float* out = a->o + b * a->os[0] + h * a->os[1] + i * a->os[2];
          const int hk = h / g;
          const int kv_len = kv_len_of(a, i);
          float maxval = -INFINITY;
          for (int j = 0; j < kv_len; j++) {          /* attn.cpp:56-69 */
            const float* k = a->k + b * a->ks[0] + hk * a->ks[1] + j * a->ks[2];
            float val = 0.f;
            for (int d = 0; d < D; d++) val += q[d] * k[d];
            val *= a->scale;
            if (val > maxval) maxval = val;
            score[j] = val;
          }
          float sum = 0.f;                            /* attn.cpp:71-80 */
          for (int j = 0; j < kv_len; j++) { float e = expf(score[j] - maxval); sum += e; score[j] = e; }
          for (int j = 0; j < kv_len; j++) score[j] /= sum;
          for (int d = 0; d < D; d++) out[d] = 0.f;   /* attn.cpp:82-95 */
          for (int j = 0; j < kv_len; j++) {
            const float* v = a->v + b * a->vs[0] + hk * a->vs[1] + j * a->vs[2];
            for (int d = 0; d < D; d++) out[d] += score[j] * v[d];
          }
          if (a->lse) a->lse[((int64_t)b * a->H + h) * a->Nq + i] = kv_len > 0 ? maxval + logf(sum) : INFINITY;
        }
    free(score);
  }
  return 0;
}

/*
 * One-pass online-softmax attention in fp32: follows run_flash_attn,
 * flash_attention_c/csrc/attn.cpp:101-169 — per key: new max, exp, rescale of the running
 * sum and of the D outputs, accumulate; divide once at the end.
 */
int oracle_attn_online(const oracle_args* a) {
  const int D = a->D, g = a->H / a->Hk;
#pragma omp parallel for collapse(3) schedule(dynamic, 8)
  for (int b = 0; b < a->B; b++)
    for (int h = 0; h < a->H; h++)
      for (int i = 0; i < a->Nq; i++) {
        const float* q = a->q + b * a->qs[0] + h * a->qs[1] + i * a->qs[2];
        float* out = a->o + b * a->os[0] + h * a->os[1] + i * a->os[2];
        const int hk = h / g;
        const int kv_len = kv_len_of(a, i);
        float maxval = -INFINITY, score_sum = 0.f;
        for (int d = 0; d < D; d++) out[d] = 0.f;
        for (int j = 0; j < kv_len; j++) {
          const float* k = a->k + b * a->ks[0] + hk * a->ks[1] + j * a->ks[2];
          float val = 0.f;
          for (int d = 0; d < D; d++) val += q[d] * k[d];   /* attn.cpp:131-134 */
          val *= a->scale;
          const float local_max = maxval > val ? maxval : val;  /* attn.cpp:137 */
          const float e = expf(val - local_max);                /* attn.cpp:141-142 */
          const float rescale = expf(maxval - local_max);
          score_sum = score_sum * rescale + e;                  /* attn.cpp:145-146 */
          const float* v = a->v + b * a->vs[0] + hk * a->vs[1] + j * a->vs[2];
          for (int d = 0; d < D; d++) out[d] = out[d] * rescale + e * v[d];  /* attn.cpp:151-155 */
          maxval = local_max;
        }
        if (kv_len > 0) {
          for (int d = 0; d < D; d++) out[d] /= score_sum;      /* attn.cpp:162-164 */
        }
        if (a->lse) a->lse[((int64_t)b * a->H + h) * a->Nq + i] = kv_len > 0 ? maxval + logf(score_sum) : INFINITY;
      }
  return 0;
}

/*
 * Ground-truth form of the GPU kernel's contract, accumulated in fp64:
 *   S = scale * q.k ; m = max S ; P = exp(S - m) ; l = sum P (unrounded)
 *   O = (sum_j round16(P_j) v_j) / l        round16 per p_round: 0 none, 1 bf16, 2 fp16
 *   LSE = m + ln l ; empty row -> O = 0, LSE = +inf
 * following flash_attention_cutlass/csrc/flash_attention.cu:263-316 (softmax_rescale_o: l sums
 * the fp32 P), :601 (P rounded to 16 bit before the second GEMM) and :608-630 (epilogue), and
 * flash_attention_py/main_torch_only.py:260 (`local_score.to(q.dtype) @ v_tile`).
 */
int oracle_attn_exact64(const oracle_args* a, int p_round) {
  const int D = a->D, g = a->H / a->Hk;
#pragma omp parallel
  {
    double* score = (double*)malloc(sizeof(double) * (size_t)(a->Nk > 0 ? a->Nk : 1));
    double* acc = (double*)malloc(sizeof(double) * (size_t)D);
#pragma omp for collapse(3) schedule(dynamic, 8)
    for (int b = 0; b < a->B; b++)
      for (int h = 0; h < a->H; h++)
        for (int i = 0; i < a->Nq; i++) {
          const float* q = a->q + b * a->qs[0] + h * a->qs[1] + i * a->qs[2];
          float* out = a->o + b * a->os[0] + h * a->os[1] + i * a->os[2];
          const int hk = h / g;
          const int kv_len = kv_len_of(a, i);
          double maxval = -INFINITY;
          for (int j = 0; j < kv_len; j++) {
            const float* k = a->k + b * a->ks[0] + hk * a->ks[1] + j * a->ks[2];
            double val = 0.0;
            for (int d = 0; d < D; d++) val += (double)q[d] * (double)k[d];
            val *= (double)a->scale;
            if (val > maxval) maxval = val;
            score[j] = val;
          }
          double sum = 0.0;
          for (int d = 0; d < D; d++) acc[d] = 0.0;
          for (int j = 0; j < kv_len; j++) {
            const double e = exp(score[j] - maxval);
            sum += e;
            const double pr = (double)round_p((float)e, p_round);
            const float* v = a->v + b * a->vs[0] + hk * a->vs[1] + j * a->vs[2];
            for (int d = 0; d < D; d++) acc[d] += pr * (double)v[d];
          }
          for (int d = 0; d < D; d++) out[d] = kv_len > 0 ? (float)(acc[d] / sum) : 0.f;
          if (a->lse) a->lse[((int64_t)b * a->H + h) * a->Nq + i] = kv_len > 0 ? (float)(maxval + log(sum)) : INFINITY;
        }
    free(score);
    free(acc);
  }
  return 0;
}

/* helpers exposed for tests */
float oracle_round_bf16(float x) { return round_bf16(x); }
float oracle_round_fp16(float x) { return round_fp16(x); }
int oracle_num_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
