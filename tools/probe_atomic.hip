// tools/probe_atomic.hip — what does accumulating dQ with fp32 atomics cost on MI355X?
// A fused backward (keys resident, dS exchanged through LDS) adds a 64 x 128 fp32 dQ tile to HBM per (key block, query tile)
// pair: at BASELINE config 3 that is ~2.2 GB of red-ops per launch into a 268 MB accumulator.  This probe issues that
// traffic pattern alone — every wave adds 16 values per lane (a 32 x 32 sub-tile, 128 contiguous bytes per row) per "tile"
// — and reports the sustained rate for: agent-scope global_atomic_add_f32, workgroup-scope, packed bf16 atomics, and plain
// 16-byte stores of the same bytes (the roof).   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o probe_atomic probe_atomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// dq: (heads, N, 128) fp32.  Workgroup (head, kb) walks query tiles qt = 0..ntile-1 of its head; wave w adds the sub-tile
// rows 32*(w&1).., columns 32*(w>>1)..
template <int MODE>
__global__ __launch_bounds__(512) void red(float* dq, int N, int ntile, int nkb) {
  const int head = blockIdx.x / nkb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = 32 * (wave & 1), d0 = 32 * (wave >> 1);
  float* base = dq + (size_t)head * N * 128;
  const float v = 1.0f;
  for (int t = 0; t < ntile; ++t) {
    const int tt = (t + blockIdx.x * 7) % ntile;          // the key blocks of a head are at different query tiles at any moment
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // MFMA C layout: lane&31 = column (here d), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
      const int row = tt * 64 + q0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float* a = base + (size_t)row * 128 + d0 + (lane & 31);
      if (MODE == 0) __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 1) __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) *a = v;                         // plain 4-byte stores, same addresses
      else if (MODE == 3) {                               // transposed ownership: lane&31 = row -> 16 contiguous floats per lane? no: 4 x 16 B
        if ((r & 3) == 0) {
          const int row2 = tt * 64 + q0 + (lane & 31);
          float4* a4 = reinterpret_cast<float4*>(base + (size_t)row2 * 128 + d0 + 8 * (r >> 2) + 4 * (lane >> 5));
          *a4 = float4{v, v, v, v};
        }
      }
    }
  }
}

int main() {
  const int heads = 128, N = 4096, nkb = 16;
  const int ntile = 32;                                   // per workgroup: 32 of the 64 query tiles (causal average)
  float* dq;
  const size_t bytes = (size_t)heads * N * 128 * 4;
  hipMalloc(&dq, bytes);
  hipMemset(dq, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double moved = (double)heads * nkb * ntile * 64 * 128 * 4;
  const char* names[4] = {"agent-scope atomic add f32", "workgroup-scope atomic add f32", "plain 4-byte stores", "plain 16-byte stores (row per lane)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(red<0>, dim3(heads * nkb), dim3(512), 0, 0, dq, N, ntile, nkb);
      if (mode == 1) hipLaunchKernelGGL(red<1>, dim3(heads * nkb), dim3(512), 0, 0, dq, N, ntile, nkb);
      if (mode == 2) hipLaunchKernelGGL(red<2>, dim3(heads * nkb), dim3(512), 0, 0, dq, N, ntile, nkb);
      if (mode == 3) hipLaunchKernelGGL(red<3>, dim3(heads * nkb), dim3(512), 0, 0, dq, N, ntile, nkb);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-40s %8.3f ms for %.2f GB = %7.1f GB/s\n", names[mode], ms, moved / 1e9, moved / ms / 1e6);
    }
  }
  // correctness of the agent-scope adds: every element was added heads-independent nkb * (ntile/64 coverage) times
  float h[4];
  hipMemcpy(h, dq, 16, hipMemcpyDeviceToHost);
  printf("dq[0] = %g\n", h[0]);
  return 0;
}
