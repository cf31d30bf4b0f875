// tools/probe_mix_power.hip — what does the attention loop's INSTRUCTION MIX sustain under the board's power cap?
// tools/probe_mfma_power.hip measured the matrix pipe alone on random bf16 operands (1.69-1.72 PF at ~1.7 GHz / 1315 W).  The forward
// loop also issues, per MFMA, one 16-byte-per-lane LDS read (the K or V fragment), about one v_exp_f32 and four plain VALU ops.
// Arms (all on normal(0,0.5) data, 2 waves per SIMD, operands rotated like the attention loop):
//   mfma              MFMAs only
//   mfma+lds          + one ds_read_b128 per MFMA, the data read IS the next A operand
//   mfma+lds/2        + one ds_read_b128 per TWO MFMAs (what a 64-row-per-wave layout would need)
//   mfma+valu         + 1 v_exp_f32 + 4 VALU per MFMA on independent registers
//   mfma+lds+valu     the attention mix
//   mfma+lds/2+valu   the attention mix with half the LDS traffic
// Prints sustained TFLOP/s per arm; run under tools/power_trace.py --cmd for power and clock.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_mix_power probe_mix_power.hip
#include <hip/hip_runtime.h>
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

template <int LDS, int VALU>      // LDS: 0 none, 1 one read per MFMA, 2 one read per two MFMAs
__global__ __launch_bounds__(512, 2) void spin(const u32x4* src, float* sink, int iters) {
  __shared__ u32x4 lds[2560];      // 40 KiB: 32 fragments of 1 KiB at a base that moves over 8 KiB
  const int tid = blockIdx.x * 512 + threadIdx.x, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2560; i += 512) lds[i] = src[(blockIdx.x * 2048 + i) & 0xfffff];
  __syncthreads();
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & 0xfffff]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 8 + i) & 0xfffff]);
  }
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[0]) + lane * 16;
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 0.01f * (float)(lane + i);
  f32x16 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    const unsigned itaddr = ldsaddr + (it & 7) * 1024;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (LDS == 1 || (LDS == 2 && (m & 1) == 0)) {      // the fragment MFMA m+4 uses; explicit asm so that hipcc cannot sink the read to its use
        u32x4 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(itaddr), "n"(1024 * m));
        a[(m + 4) & 7] = __builtin_bit_cast(bf16x8, t);
        if (LDS == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[m & 7]));
        else          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[m & 7]));
      }
      c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m / 2) & 7], c[m & 3], 0, 0, 0);
      if (VALU) {
        float& y = x[m & 7];
        y = __builtin_amdgcn_exp2f(__builtin_fmaf(y, -0.731f, 0.25f));      // 1 fma + 1 exp
        y = __builtin_fmaf(y, 1.37f, x[(m + 3) & 7]);                        // + 3 more VALU ops
        y = __builtin_fmaxf(y, -4.f);
        y = __builtin_fminf(y, 4.f);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[k][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) sink[tid] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// the same mixes on v_mfma_f32_16x16x32_bf16: TWO MFMAs (2 x 16 KFLOP) carry what ONE 32x32x16 carried — one LDS read whose data
// feeds both (the K / V fragment serves two 16-row blocks), 1 exp + 4 VALU twice... no: the companions are per flop, so once per pair
template <int LDS, int VALU>
__global__ __launch_bounds__(512, 2) void spin16(const u32x4* src, float* sink, int iters) {
  __shared__ u32x4 lds[2560];
  const int tid = blockIdx.x * 512 + threadIdx.x, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2560; i += 512) lds[i] = src[(blockIdx.x * 2048 + i) & 0xfffff];
  __syncthreads();
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & 0xfffff]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 8 + i) & 0xfffff]);
  }
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[0]) + lane * 16;
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 0.01f * (float)(lane + i);
  f32x4 c[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[k][r] = 0.f;
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    const unsigned itaddr = ldsaddr + (it & 7) * 1024;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (LDS == 1 || (LDS == 2 && (m & 1) == 0)) {
        u32x4 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(itaddr), "n"(1024 * m));
        a[(m + 4) & 7] = __builtin_bit_cast(bf16x8, t);
        if (LDS == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[m & 7]));
        else          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[m & 7]));
      }
      c[(2 * m) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 7], b[(m / 2) & 7], c[(2 * m) & 15], 0, 0, 0);
      c[(2 * m + 1) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 7], b[(m / 2 + 4) & 7], c[(2 * m + 1) & 15], 0, 0, 0);
      if (VALU) {
        float& y = x[m & 7];
        y = __builtin_amdgcn_exp2f(__builtin_fmaf(y, -0.731f, 0.25f));
        y = __builtin_fmaf(y, 1.37f, x[(m + 3) & 7]);
        y = __builtin_fmaxf(y, -4.f);
        y = __builtin_fminf(y, 4.f);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += c[k][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) sink[tid] = s;
}

static float gauss() {
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const size_t n16 = (size_t)(1 << 20) * 8;
  std::vector<unsigned short> h(n16);
  u32x4* src;
  float* sink;
  hipMalloc(&src, n16 * 2);
  hipMalloc(&sink, 1024 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 1024, iters = 4000;
  const double flops = (double)grid * 8 * iters * 32 * 32768.0;
  for (size_t i = 0; i < n16; ++i) {
    float x = 0.5f * gauss();
    unsigned u; memcpy(&u, &x, 4);
    h[i] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
  }
  hipMemcpy(src, h.data(), n16 * 2, hipMemcpyHostToDevice);
  const char* names[9] = {"mfma", "mfma+lds", "mfma+lds/2", "mfma+valu", "mfma+lds+valu", "mfma+lds/2+valu",
                          "16x16x32:mfma", "16x16x32:mfma+lds+valu", "16x16x32:mfma+lds/2+valu"};
  for (int arm = 0; arm < 9; ++arm) {
    std::this_thread::sleep_for(std::chrono::milliseconds(700));
    printf("ARM_BEGIN %s\n", names[arm]); fflush(stdout);
    const auto t0 = std::chrono::steady_clock::now();
    double best = 0, last = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
      hipEventRecord(e0);
      for (int rep = 0; rep < 4; ++rep) {
        if (arm == 0) hipLaunchKernelGGL((spin<0, 0>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 1) hipLaunchKernelGGL((spin<1, 0>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 2) hipLaunchKernelGGL((spin<2, 0>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 3) hipLaunchKernelGGL((spin<0, 1>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 4) hipLaunchKernelGGL((spin<1, 1>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 5) hipLaunchKernelGGL((spin<2, 1>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 6) hipLaunchKernelGGL((spin16<0, 0>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 7) hipLaunchKernelGGL((spin16<1, 1>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        if (arm == 8) hipLaunchKernelGGL((spin16<2, 1>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      last = 4 * flops / (ms * 1e-3) / 1e12;
      if (last > best) best = last;
    }
    printf("ARM_END %s  sustained %.1f TFLOP/s (last batch; best %.1f)\n", names[arm], last, best);
    fflush(stdout);
  }
  return 0;
}
