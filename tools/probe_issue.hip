// tools/probe_issue.hip — how many independent VALU ops fit under one v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles)
//  (a) issued by the SAME wave between consecutive MFMAs, (b) issued by ANOTHER wave of the same SIMD while the MFMA wave
//  pads its stream with s_nop.  All instruction streams are volatile inline asm so the order is exactly as written.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
// OP: 0 v_fma_f32, 1 v_exp_f32, 2 v_pk_mul_f32, 3 v_max_f32, 4 v_cvt_pk_bf16_f32
template <int OP> __device__ __forceinline__ void valu(float& x, f32x2& p, float s) {
  if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
  if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(s));
  if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(s));
  if (OP == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(x));
  if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  if (OP == 7) asm volatile("v_dot2_f32_bf16 %0, %1, %1, %0" : "+v"(x) : "v"(s));
  if (OP == 8) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(s));
  if (OP == 9) asm volatile("v_exp_bf16 %0, %0" : "+v"(x));
}

// ROLE 1: MFMA with K VALU ops of kind OP after every MFMA and NOP s_nop-8-cycle pads; ROLE 2: VALU only (12 per "slot"); 0 idle
template <int K, int OP, int NOPS>
__device__ __forceinline__ float role_mfma(int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float x[12]; f32x2 p[12];
  for (int e = 0; e < 12; ++e) { x[e] = lane * 0.001f + e; p[e] = f32x2{x[e], 1.f}; }
  float s = 0.999f;
  for (int i = 0; i < iters; ++i) {
#define SLOT(c)                                                   \
    MFMA(c);                                                      \
    _Pragma("unroll") for (int e = 0; e < K; ++e) valu<OP>(x[e % 12], p[e % 12], s); \
    _Pragma("unroll") for (int e = 0; e < NOPS; ++e) asm volatile("s_nop 7");
    SLOT(c0) SLOT(c1) SLOT(c2) SLOT(c3)
  }
  float r = c0[0] + c1[1] + c2[2] + c3[3];
  for (int e = 0; e < 12; ++e) r += x[e] + p[e][0];
  return r;
}
template <int OP> __device__ __forceinline__ float role_valu(int iters) {
  const int lane = threadIdx.x & 63;
  float x[12]; f32x2 p[12];
  for (int e = 0; e < 12; ++e) { x[e] = lane * 0.001f + e; p[e] = f32x2{x[e], 1.f}; }
  float s = 0.999f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 24; ++e) valu<OP>(x[e % 12], p[e % 12], s);
  }
  float r = 0;
  for (int e = 0; e < 12; ++e) r += x[e] + p[e][0];
  return r;
}

template <int K, int OP, int NOPS, int BROLE>   // BROLE: 0 idle, 1 same as A, 2 VALU-only(24 ops/iter)
__global__ __launch_bounds__(512, 2) void k(float* out, long long* tr, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  float r = 0;
  if (wave < 4) r = role_mfma<K, OP, NOPS>(iters);
  else if (BROLE == 1) r = role_mfma<K, OP, NOPS>(iters);
  else if (BROLE == 2) r = role_valu<OP>(iters);
  asm volatile("" ::"v"(r));
  long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { tr[wave * 2] = t0; tr[wave * 2 + 1] = t1; }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

static const char* opn[] = {"v_fma_f32", "v_exp_f32", "v_pk_mul_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "v_exp_f16", "v_pk_add_f32", "v_dot2_f32_bf16", "v_pk_mul_f16", "v_exp_bf16"};
template <int K, int OP, int NOPS, int BROLE> void run(float* d, long long* tr) {
  const int it = 10000;
  k<K, OP, NOPS, BROLE><<<256, 512>>>(d, tr, it);
  hipDeviceSynchronize();
  k<K, OP, NOPS, BROLE><<<256, 512>>>(d, tr, it);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, tr, sizeof(h), hipMemcpyDeviceToHost);
  printf("K=%2d %-18s nops=%d B=%s : waveA %6.1f cycles per MFMA-slot", K, opn[OP], NOPS, BROLE == 0 ? "idle" : BROLE == 1 ? "same" : "VALU",
         double(h[1] - h[0]) / it / 4);
  if (BROLE) printf(", waveB done after %6.1f cycles per slot (B issues %s)", double(h[9] - h[8]) / it / 4, BROLE == 2 ? "6 VALU per slot" : "same stream");
  printf("\n");
}

int main() {
  float* d; hipMalloc(&d, 4096);
  long long* tr; hipMalloc(&tr, 4096);
  printf("== (a) same wave: K independent VALU ops after each MFMA (32 cycles matrix pipe per MFMA), one wave per SIMD\n");
  run<0, 0, 0, 0>(d, tr); run<2, 0, 0, 0>(d, tr); run<4, 0, 0, 0>(d, tr); run<6, 0, 0, 0>(d, tr); run<8, 0, 0, 0>(d, tr); run<10, 0, 0, 0>(d, tr); run<12, 0, 0, 0>(d, tr);
  run<2, 1, 0, 0>(d, tr); run<4, 1, 0, 0>(d, tr); run<6, 1, 0, 0>(d, tr); run<8, 1, 0, 0>(d, tr);
  run<4, 2, 0, 0>(d, tr); run<8, 2, 0, 0>(d, tr); run<12, 2, 0, 0>(d, tr);
  run<4, 3, 0, 0>(d, tr); run<8, 3, 0, 0>(d, tr);
  run<4, 4, 0, 0>(d, tr); run<8, 4, 0, 0>(d, tr);
  run<4, 5, 0, 0>(d, tr); run<8, 5, 0, 0>(d, tr);
  run<4, 6, 0, 0>(d, tr); run<8, 6, 0, 0>(d, tr);
  run<4, 7, 0, 0>(d, tr); run<8, 7, 0, 0>(d, tr);
  run<4, 8, 0, 0>(d, tr); run<8, 8, 0, 0>(d, tr);
#ifdef HAVE_EXP_BF16
  run<4, 9, 0, 0>(d, tr); run<8, 9, 0, 0>(d, tr);
#endif
  printf("== (a2) two such waves per SIMD\n");
  run<6, 0, 0, 1>(d, tr); run<4, 5, 0, 1>(d, tr); run<8, 5, 0, 1>(d, tr); run<8, 7, 0, 1>(d, tr); run<8, 6, 0, 1>(d, tr);
  run<0, 0, 0, 1>(d, tr); run<4, 0, 0, 1>(d, tr); run<8, 0, 0, 1>(d, tr); run<12, 0, 0, 1>(d, tr);
  run<4, 1, 0, 1>(d, tr); run<8, 1, 0, 1>(d, tr);
  printf("== (b) other wave issues VALU (6 per slot); MFMA wave pads with s_nop 7 (8 cycles each)\n");
  run<0, 0, 0, 2>(d, tr); run<0, 0, 1, 2>(d, tr); run<0, 0, 2, 2>(d, tr); run<0, 0, 3, 2>(d, tr);
  run<0, 1, 0, 2>(d, tr); run<0, 1, 3, 2>(d, tr);
  return 0;
}
