#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(512, 2) void v_mine(const u32x4* src, unsigned chunk_mask, float* sink, int iters) {
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

__global__ __launch_bounds__(512, 2) void v_constmask(const u32x4* src, unsigned chunk_mask, float* sink, int iters) {
  const unsigned tid = blockIdx.x * 512 + threadIdx.x;
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
  if (s == 123.456f && sink) sink[tid] = s;
}

__global__ __launch_bounds__(512, 2) void v_inttid(const u32x4* src, unsigned chunk_mask, float* sink, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
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

__global__ __launch_bounds__(512, 2) void v_nosinkchk(const u32x4* src, unsigned chunk_mask, float* sink, int iters) {
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
  if (s == 123.456f ) sink[tid] = s;
}

__global__ __launch_bounds__(512, 2) void v_theirs(const u32x4* src, unsigned chunk_mask, float* sink, int iters) {
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
  if (s == 123.456f ) sink[tid] = s;
}
typedef void (*kern_t)(const u32x4*, unsigned, float*, int);
int main() {
  u32x4* src; float* sink;
  hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20); hipMalloc(&sink, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char* name; kern_t k; } ks[] = {{"v_mine", v_mine}, {"v_constmask", v_constmask}, {"v_inttid", v_inttid}, {"v_nosinkchk", v_nosinkchk}, {"v_theirs", v_theirs}};
  const int iters = 2000; const double flops = 1024.0 * 8 * iters * 32 * 32768.0 * 4;
  for (int rep = 0; rep < 2; ++rep)
  for (auto& kk : ks) {
    double last = 0; const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) {
      hipEventRecord(e0);
      for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kk.k, dim3(1024), dim3(512), 0, 0, src, 0xfffffu, sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); last = flops / (ms * 1e-3) / 1e12;
    }
    printf("zeros %-14s %7.1f TFLOP/s\n", kk.name, last); fflush(stdout);
  }
  return 0;
}
