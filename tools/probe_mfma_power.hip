// tools/probe_mfma_power.hip — what does the matrix pipe itself sustain on RANDOM operands under the board's power cap?
// The forward kernel runs at the 1400 W cap on the reference's normal(0,0.5) inputs (profiles/r03_power_trace_cfg3.txt), so
// 2.5 PF (the peak at 2.4 GHz) is not reachable on that data by ANY instruction stream.  This probe measures the ceiling:
// nothing but v_mfma_f32_32x32x16_bf16, two waves per SIMD, operands held in registers and ROTATED the way the attention
// loop rotates them (a new A fragment every MFMA, a new B fragment every `breuse` MFMAs, four accumulators round-robin).
// Arms: operand values normal(0,0.5) / zeros; B reuse 1, 2, 4.  Prints TFLOP/s per arm; run it under tools/power_trace.py --cmd
// for the power and clock of each arm.   build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma_power probe_mfma_power.hip
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <chrono>
#include <thread>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int BREUSE>
__global__ __launch_bounds__(512, 2) void spin(const u32x4* src, float* sink, int iters) {
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
    for (int m = 0; m < 32; ++m)       // 32 MFMAs per iteration: A rotates every MFMA, B every BREUSE MFMAs, accumulators round-robin
      c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m / BREUSE) & 7], c[m & 3], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[k][r];
  if (s == 123.456f) sink[tid] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// the same stream with v_mfma_f32_16x16x32_bf16 (16 KFLOP per instruction, a quarter of the accumulator registers per MFMA):
// does the other tile shape sustain more under the cap?  16 accumulators round-robin, 64 MFMAs per iteration = the same flops.
template <int BREUSE>
__global__ __launch_bounds__(512, 2) void spin16(const u32x4* src, float* sink, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & 0xfffff]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 8 + i) & 0xfffff]);
  }
  f32x4 c[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[k][r] = 0.f;
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m)
      c[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 7], b[(m / BREUSE) & 7], c[m & 15], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += c[k][r];
  if (s == 123.456f) sink[tid] = s;
}

static float gauss() {
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const size_t n16 = (size_t)(1 << 20) * 8;          // 1M 16-byte chunks
  std::vector<unsigned short> h(n16);
  u32x4* src;
  float* sink;
  hipMalloc(&src, n16 * 2);
  hipMalloc(&sink, 1024 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 1024, iters = 4000;
  const double flops = (double)grid * 8 * iters * 32 * 32768.0;
  for (int data = 0; data < 2; ++data) {
    for (size_t i = 0; i < n16; ++i) {
      float x = data == 0 ? 0.5f * gauss() : 0.f;
      unsigned u; memcpy(&u, &x, 4);
      h[i] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
    hipMemcpy(src, h.data(), n16 * 2, hipMemcpyHostToDevice);
    for (int br = 0; br < 4; ++br) {
      const int breuse = br < 3 ? 1 << br : 2;
      const bool shape16 = br == 3;                // fourth arm: the 16x16x32 instruction, B reuse 2
      // idle gap so that a power sampler can tell the arms apart
      std::this_thread::sleep_for(std::chrono::milliseconds(700));
      printf("ARM_BEGIN data=%s breuse=%d%s\n", data == 0 ? "normal(0,0.5)" : "zeros", breuse, shape16 ? " 16x16x32" : ""); fflush(stdout);
      const auto t0 = std::chrono::steady_clock::now();
      double best = 0, last = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int rep = 0; rep < 4; ++rep) {
          if (shape16) { hipLaunchKernelGGL(spin16<2>, dim3(grid), dim3(512), 0, 0, src, sink, iters); continue; }
          if (breuse == 1) hipLaunchKernelGGL(spin<1>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
          if (breuse == 2) hipLaunchKernelGGL(spin<2>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
          if (breuse == 4) hipLaunchKernelGGL(spin<4>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        last = 4 * flops / (ms * 1e-3) / 1e12;
        if (last > best) best = last;
      }
      printf("ARM_END data=%s breuse=%d%s  sustained %.1f TFLOP/s (last batch; best %.1f)\n", data == 0 ? "normal(0,0.5)" : "zeros", breuse, shape16 ? " 16x16x32" : "", last, best);
      fflush(stdout);
    }
  }
  return 0;
}
