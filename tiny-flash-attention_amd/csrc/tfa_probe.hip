// tfa_probe.hip — tfa_debug_mfma_ceiling: what nothing but v_mfma_f32_32x32x16_bf16 sustains on THIS box, on the caller's operand values.
// The forward kernel runs at the board's power cap on the reference's normal(0, 0.5) inputs, so the nominal 2.5 PF (a 2.4 GHz figure) is not
// reachable on that data by any instruction stream; the sustained rate of a bare MFMA stream varies by +-5 % between boxes of the pool
// (docs/LABLOG.md L-6: 1.69-1.85 PF).  bench.py measures it in the same run, on the same q tensor, and quotes it BESIDE the nominal peak
// (`roofline.mfma_only_ceiling_random_data`) — never instead of it.  Same stream as tools/probe_mfma_power.hip: two waves per SIMD, operands
// in registers, a new A fragment every MFMA, a new B fragment every second MFMA, four accumulators round-robin (how the attention loop rotates them).
#include <hip/hip_runtime.h>
#include <chrono>
#include <algorithm>
#include "tfa.h"

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// (the operand index mask is a CONSTANT — the first 16 MiB of the buffer — on purpose: with a run-time mask hipcc forms 64-bit vector addresses for the sixteen
//  loads in front of the loop, and that build of the very same loop runs 20 % slower, on zeros as on random data: 1.98 vs 2.47 PF, tools/probe_mfma_variants.hip;
//  neither the operand registers hipcc picks nor the loop's alignment explain it: tools/probe_mfma_regs.hip, probe_mfma_align.hip, docs/LABLOG.md L-11)
__global__ __launch_bounds__(512, 2) void mfma_stream(const u32x4* src, float* sink, int iters) {
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
  if (s == 123.456f) sink[tid] = s;              // (never true for finite data: keeps the accumulators alive)
}
}  // namespace

extern "C" int tfa_debug_mfma_ceiling(const void* operands, unsigned long long bytes, double seconds, void* stream, double* tflops) {
  if (!operands || !tflops) return TFA_ERR_NULL;
  *tflops = 0.0;
  if (bytes < (16ull << 20) || !(seconds > 0.0) || seconds > 30.0) return TFA_ERR_SHAPE;   // the kernel reads the first 16 MiB
  // (the sink is written only if a lane's sum equals a magic number — never for finite data.  One per call, on the device that is current NOW: a static
  //  pointer would belong to whichever device and thread came first.  The call synchronises the stream, so it cannot be part of a stream capture: tfa.h)
  float* sink = nullptr;
  if (hipMalloc(&sink, 1024 * 512 * sizeof(float)) != hipSuccess) return (int)hipGetLastError();
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess) { const int rc = (int)hipGetLastError(); (void)hipFree(sink); return rc; }
  if (hipEventCreate(&e1) != hipSuccess) { const int rc = (int)hipGetLastError(); (void)hipEventDestroy(e0); (void)hipFree(sink); return rc; }
  const int grid = 1024, iters = 4000, reps = 2;   // ~40 ms per group: host round trips between groups stay below 0.3 %
  const double flops = (double)grid * 8 * iters * 32 * 32768.0 * reps;
  double rates[512];
  int nr = 0;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = TFA_OK;
  do {                                             // groups of `reps` launches between two events; the median of the second half of the groups is reported
    (void)hipGetLastError();
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_stream, dim3(grid), dim3(512), 0, s, (const u32x4*)operands, sink, iters);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) { rc = (int)hipGetLastError(); break; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms > 0.f && nr < 512) rates[nr++] = flops / (ms * 1e-3) / 1e12;
  } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (rc != TFA_OK) return rc;
  if (nr == 0) return (int)hipErrorNotReady;       // not one timed group: nothing to report (*tflops stays 0; positive = a HIP error code, tfa.h)
  {                                                // (the first half of the groups is the clock settling on this stream: the firmware's power loop swings +-15 % per group)
    const int lo = nr / 2, n = nr - lo;
    for (int i = lo + 1; i < nr; ++i) {            // insertion sort of the second half
      const double x = rates[i];
      int j = i - 1;
      while (j >= lo && rates[j] > x) { rates[j + 1] = rates[j]; --j; }
      rates[j + 1] = x;
    }
    *tflops = rates[lo + n / 2];
  }
  return rc;
}
