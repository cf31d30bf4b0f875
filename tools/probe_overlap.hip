// tools/probe_overlap.hip — does a matrix-only wave overlap with a VALU-only (or LDS-only) wave on the SAME SIMD?
// 512-thread workgroups, 1 per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 (same SIMDs) run role B.
// roles: 0 idle, 1 MFMA chain (4 accumulators), 2 VALU fma+exp, 3 LDS ds_read_b128, 4 MFMA with 3 VALU between MFMAs,
//        5 VALU fma only (no transcendental), 6 MFMA 16x16x32
// PRIO: 0 none, 1 role-A waves s_setprio 3, 2 role-B waves s_setprio 3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, char* lds) {
  const int lane = threadIdx.x & 63;
  float acc_out = 0.f;
  if (ROLE == 1 || ROLE == 4) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float x0 = lane * 0.001f, x1 = 0.5f, x2 = 0.25f;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      if (ROLE == 4) { x0 = __builtin_amdgcn_exp2f(x0 * 0.5f - 1.f); x1 = fmaf(x1, 0.99f, x0); x2 += x1; }
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      if (ROLE == 4) { x0 = __builtin_amdgcn_exp2f(x0 * 0.5f - 1.f); x1 = fmaf(x1, 0.99f, x0); x2 += x1; }
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      if (ROLE == 4) { x0 = __builtin_amdgcn_exp2f(x0 * 0.5f - 1.f); x1 = fmaf(x1, 0.99f, x0); x2 += x1; }
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      if (ROLE == 4) { x0 = __builtin_amdgcn_exp2f(x0 * 0.5f - 1.f); x1 = fmaf(x1, 0.99f, x0); x2 += x1; }
    }
    acc_out = c0[0] + c1[1] + c2[2] + c3[3] + x2;
  } else if (ROLE == 6) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    f32x4 c[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int e = 0; e < 8; ++e) c[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[e], 0, 0, 0);
    }
    for (int e = 0; e < 8; ++e) acc_out += c[e][e & 3];
  } else if (ROLE == 2 || ROLE == 5) {
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = lane * 0.001f + e;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {           // per element 4 VALU (role 2: one of them transcendental)
        float t = fmaf(x[e], 0.7f, -0.3f);
        if (ROLE == 2) t = __builtin_amdgcn_exp2f(t); else t = fmaf(t, t, 0.125f);
        x[e] = fmaf(t, 0.5f, x[e] * 0.25f);
      }
    }
    for (int e = 0; e < 8; ++e) acc_out += x[e];
  } else if (ROLE == 3) {
    u32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        u32x4 v = *reinterpret_cast<const volatile u32x4*>(lds + ((lane * 16 + e * 1024 + (i & 3) * 8192) & 65535));
        s[0] += v[0]; s[1] ^= v[1]; s[2] += v[2]; s[3] ^= v[3];
      }
    }
    acc_out = (float)(s[0] + s[1] + s[2] + s[3]);
  }
  return acc_out;
}

template <int RA, int RB, int PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* tr, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (PRIO == 1 && wave < 4) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 2 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  long long t0 = __builtin_readcyclecounter();
  float r = (wave < 4) ? run_role<RA>(iters, lds) : run_role<RB>(iters, lds);
  asm volatile("" ::"v"(r));
  long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { tr[wave * 2] = t0; tr[wave * 2 + 1] = t1; }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int RA, int RB, int PRIO> void timeit(float* d, long long* tr, int iters, const char* name) {
  hipFuncSetAttribute((const void*)k<RA, RB, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<RA, RB, PRIO><<<256, 512, 65536>>>(d, tr, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<RA, RB, PRIO><<<256, 512, 65536>>>(d, tr, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  long long h[16]; hipMemcpy(h, tr, sizeof(h), hipMemcpyDeviceToHost);
  long long base = h[0];
  double dA = double(h[1] - h[0]) / iters, dB = double(h[9] - h[8]) / iters;
  double mhz = double(h[1] - h[0] > h[9] - h[8] ? h[1] - h[0] : h[9] - h[8]) / (ms * 1e3);
  printf("%-44s %7.3f ms | wave0 %7.1f ticks/iter, wave4 %7.1f ticks/iter (s_memtime ticks; longest/ms = %.0f MHz) err=%s\n", name, ms, dA, dB,
         mhz, hipGetErrorString(hipGetLastError()));
  (void)base;
}

int main() {
  float* d; hipMalloc(&d, 4096);
  long long* tr; hipMalloc(&tr, 4096);
  const int it = 20000;
  printf("iters=%d per wave; MFMA role = 4 MFMA 32x32x16/iter (128 matrix-pipe cycles), VALU role = 32 VALU/iter, LDS role = 8 ds_read_b128/iter\n", it);
#define T(A, B, P, name) timeit<A, B, P>(d, tr, it, name);
  T(1, 0, 0, "A=MFMA  B=idle");
  T(1, 1, 0, "A=MFMA  B=MFMA");
  T(6, 0, 0, "A=MFMA16x16x32(8/iter)  B=idle");
  T(6, 6, 0, "A=MFMA16  B=MFMA16");
  T(2, 0, 0, "A=VALU(exp)  B=idle");
  T(2, 2, 0, "A=VALU(exp)  B=VALU(exp)");
  T(5, 0, 0, "A=VALU(fma)  B=idle");
  T(5, 5, 0, "A=VALU(fma)  B=VALU(fma)");
  T(1, 2, 0, "A=MFMA  B=VALU(exp)");
  T(1, 2, 1, "A=MFMA(prio3)  B=VALU(exp)");
  T(1, 2, 2, "A=MFMA  B=VALU(exp)(prio3)");
  T(1, 5, 0, "A=MFMA  B=VALU(fma)");
  T(1, 5, 1, "A=MFMA(prio3)  B=VALU(fma)");
  T(6, 5, 0, "A=MFMA16  B=VALU(fma)");
  T(3, 0, 0, "A=LDS  B=idle");
  T(1, 3, 0, "A=MFMA  B=LDS");
  T(2, 3, 0, "A=VALU(exp)  B=LDS");
  T(4, 0, 0, "A=MFMA+3VALU/mfma  B=idle");
  T(4, 4, 0, "A=MFMA+3VALU/mfma  B=same");
  T(4, 2, 0, "A=MFMA+3VALU/mfma  B=VALU(exp)");
  T(4, 2, 1, "A=MFMA+3VALU/mfma(prio3)  B=VALU(exp)");
  return 0;
}
