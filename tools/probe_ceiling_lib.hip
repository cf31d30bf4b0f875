// tools/probe_ceiling_lib.hip — tfa_debug_mfma_ceiling called from a plain HIP host program (no torch) next to a LOCAL copy of the same kernel timed the way
// tools/probe_mfma_power.hip times it: does the in-library MFMA-only probe measure the same?
// build: hipcc --offload-arch=gfx950 -O3 -Iinclude -o tools/probe_ceiling_lib tools/probe_ceiling_lib.hip -Ltiny-flash-attention_amd/lib -ltfa_hip -Wl,-rpath,$PWD/tiny-flash-attention_amd/lib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <chrono>
#include <time.h>
#include "tfa.h"
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(512, 2) void local_stream(const u32x4* src, unsigned chunk_mask, float* sink, int iters) {
  const unsigned tid = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & chunk_mask]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 8 + i) & chunk_mask]);
  }
  f32x16 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m / 2) & 7], c[m & 3], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[k][r];
  if (s == 123.456f && sink) sink[tid] = s;
}
// tools/probe_mfma_power.hip's kernel, verbatim (BREUSE = 2)
__global__ __launch_bounds__(512, 2) void spin2(const u32x4* src, float* sink, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & 0xfffff]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 8 + i) & 0xfffff]);
  }
  f32x16 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m / 2) & 7], c[m & 3], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[k][r];
  if (s == 123.456f) sink[tid] = s;
}
static float* g_sink = nullptr;
static void time_theirs(const void* src, int iters, int reps, double seconds, const char* tag, int sleep_ms) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  if (sleep_ms) { struct timespec ts = {0, sleep_ms * 1000000L}; nanosleep(&ts, nullptr); }
  const double flops = 1024.0 * 8 * iters * 32 * 32768.0 * reps;
  const auto t0 = std::chrono::steady_clock::now();
  double last = 0, first = 0; int n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(spin2, dim3(1024), dim3(512), 0, 0, (const u32x4*)src, g_sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    last = flops / (ms * 1e-3) / 1e12; if (n++ == 0) first = last;
  }
  printf("  THEIR kernel, %s, idle gap %d ms: first %.1f, last %.1f TFLOP/s (%d groups)\n", tag, sleep_ms, first, last, n); fflush(stdout);
}
static float gauss() {
  const float u1 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}
static void time_local(const void* src, unsigned mask, int iters, int reps, double seconds, const char* tag) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double flops = 1024.0 * 8 * iters * 32 * 32768.0 * reps;
  const auto t0 = std::chrono::steady_clock::now();
  double last = 0, first = 0; int n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(local_stream, dim3(1024), dim3(512), 0, 0, (const u32x4*)src, mask, (float*)nullptr, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    last = flops / (ms * 1e-3) / 1e12; if (n++ == 0) first = last;
  }
  printf("  local copy, %s: iters %d x %d launches per group: first %.1f, last %.1f TFLOP/s (%d groups)\n", tag, iters, reps, first, last, n); fflush(stdout);
}
int main(int argc, char** argv) {
  const size_t n16 = (size_t)(1 << 20) * 8;
  std::vector<unsigned short> h(n16);
  void* src;
  hipMalloc(&src, n16 * 2);
  hipMalloc(&g_sink, 1024 * 512 * 4);
  for (int data = 0; data < 2; ++data) {
    for (size_t i = 0; i < n16; ++i) {
      float x = data == 0 ? 0.5f * gauss() : 0.f;
      unsigned u; memcpy(&u, &x, 4);
      h[i] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
    hipMemcpy(src, h.data(), n16 * 2, hipMemcpyHostToDevice);
    const char* tag = data == 0 ? "normal(0,0.5)" : "zeros";
    for (int rep = 0; rep < 2; ++rep) {
      double tf = 0;
      int st = tfa_debug_mfma_ceiling(src, n16 * 2, 2.0, nullptr, &tf);
      printf("library, %s: status %d, %.1f TFLOP/s\n", tag, st, tf); fflush(stdout);
      time_local(src, 0xfffff, 4000, 4, 2.0, tag);
      time_theirs(src, 4000, 4, 2.0, tag, 0);
      time_theirs(src, 4000, 4, 2.0, tag, 700);
      time_local(src, 0xfffff, 4000, 4, 2.0, tag);
    }
  }
  return 0;
}
