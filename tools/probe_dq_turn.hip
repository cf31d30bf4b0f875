// tools/probe_dq_turn.hip — sibling of probe_atomic.hip: what does accumulating dQ with PLAIN load-add-store, ordered by a turn counter, cost?
// A fused 5-unit backward (dQ folded into the dK/dV launch) adds one 64 x 128 fp32 dQ tile per (key block, query tile) pair to a per-head
// accumulator (2 MB per head at N = 4096) that the key blocks of the head share.  Deterministic form: a counter per (head, query tile); key block
// kb may add only when the counter equals its rank among the tile's contributors, then bumps it.  Key blocks of a head sit on one XCD (block id
// & 7, as bwd_kv_kernel places them), so the accumulator can live in that XCD's L2.
// This probe issues THAT traffic alone (no GEMMs): BASELINE config 3's pairs (128 heads, N 4096, 256-key blocks, causal: block kb meets query
// tiles 4 kb .. 63), every workgroup walking its tiles from the LAST one down, so that it waits only for workgroups with a lower id (dispatched
// earlier: no deadlock whatever the residency).   modes:
//   0  ordered, agent-scope acquire/release around the tile (what could ship: correct wherever the two workgroups run)
//   1  ordered, same-XCD protocol by hand: sc1 (agent scope: L1-bypassing, served by the XCD's L2) loads, stores complete (vmcnt 0) before the counter store; relies on the placement
//   2  unordered plain load-add-store (racy: the roof of the traffic pattern)
//   3  agent-scope fp32 atomic adds (probe_atomic.hip's number, same pairs)
//   5  as 1 with the delay of mode 4
//   4  as 0 with a delay of ~D microseconds of ALU work per tile in front of the add (the GEMM time of a tile step) — does the hand-off hide?
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o probe_dq_turn probe_dq_turn.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int TILE_F4 = 64 * 128 / 4;       // float4s per dQ tile

template <int MODE>
__global__ __launch_bounds__(512) void acc(float* dq, unsigned* turn, int N, int nkb, int spin_cycles, unsigned* xcc) {
  if (threadIdx.x == 0) { unsigned id_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id_)); xcc[blockIdx.x] = id_ & 15; }
  const int id = blockIdx.x, x = id & 7, q8 = id >> 3;
  const int head = x + 8 * (q8 / nkb), kb = q8 % nkb;
  const int ntile = N / 64, per_kb = ntile / nkb;
  const int t_begin = kb * per_kb;                       // causal: the first query tile that sees this key block
  float4* const base = reinterpret_cast<float4*>(dq + (size_t)head * N * 128);
  unsigned* const my_turn = turn + head * ntile;
  const float4 v = float4{1.f, 1.f, 1.f, 1.f};
  for (int qt = ntile - 1; qt >= t_begin; --qt) {
    if ((MODE == 4 || MODE == 5) && spin_cycles > 0) {                  // stand-in for the tile's GEMMs
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_cycles) __builtin_amdgcn_s_sleep(8);
    }
    float4* const tile = base + (size_t)qt * TILE_F4;
    if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) {
      if (threadIdx.x == 0) {
        while (__hip_atomic_load(my_turn + qt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)kb) __builtin_amdgcn_s_sleep(2);
      }
      __syncthreads();
      if (MODE != 1 && MODE != 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (MODE == 3) {
      float* f = reinterpret_cast<float*>(tile);
#pragma unroll
      for (int i = 0; i < 16; ++i) __hip_atomic_fetch_add(f + threadIdx.x + 512 * i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 1 || MODE == 5) {
      float4 a[4];
      const float4* t0 = tile + threadIdx.x;                // (one statement: the loads AND their wait, so the adds cannot move in front of the wait)
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                   "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "v"(t0), "v"(t0 + 512), "v"(t0 + 1024), "v"(t0 + 1536) : "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i].x += v.x; a[i].y += v.y; a[i].z += v.z; a[i].w += v.w; tile[threadIdx.x + 512 * i] = a[i]; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      float4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tile[threadIdx.x + 512 * i];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i].x += v.x; a[i].y += v.y; a[i].z += v.z; a[i].w += v.w; tile[threadIdx.x + 512 * i] = a[i]; }
    }
    if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) {
      if (MODE != 1 && MODE != 5) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(my_turn + qt, (unsigned)kb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main(int argc, char** argv) {
  const int heads = 128, N = 4096, nkb = 16, ntile = N / 64;
  const int LDS = 100 * 1024;                              // one workgroup per CU, as the backward kernels run
  float* dq;
  unsigned* turn;
  const size_t bytes = (size_t)heads * N * 128 * 4;
  hipMalloc(&dq, bytes);
  hipMalloc(&turn, heads * ntile * 4);
  hipFuncSetAttribute((const void*)acc<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipFuncSetAttribute((const void*)acc<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipFuncSetAttribute((const void*)acc<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipFuncSetAttribute((const void*)acc<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipFuncSetAttribute((const void*)acc<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  long long pairs = 0;
  for (int kb = 0; kb < nkb; ++kb) pairs += ntile - kb * (ntile / nkb);
  const double moved = (double)heads * pairs * 64 * 128 * 4;     // bytes ADDED (the load-add-store forms move twice that through L2)
  unsigned* xcc;
  hipMalloc(&xcc, heads * nkb * 4);
  hipFuncSetAttribute((const void*)acc<5>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const char* names[10] = {"ordered, agent-scope fences", "ordered, same-XCD protocol (sc1 loads)", "unordered load-add-store (racy roof)", "agent-scope atomic add f32",
                          "ordered (agent), 2 us of other work per tile", "ordered (agent), 4 us of other work per tile", "ordered (agent), 6 us of other work per tile",
                          "ordered (same XCD), 2 us of other work per tile", "ordered (same XCD), 4 us of other work per tile", "ordered (same XCD), 6 us of other work per tile"};
  float* h = (float*)malloc(bytes);
  for (int mode = 0; mode < 10; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(dq, 0, bytes);
      hipMemset(turn, 0, heads * ntile * 4);
      hipDeviceSynchronize();
      const int cyc = mode >= 7 ? (mode - 6) * 200 : mode >= 4 ? (mode - 3) * 200 : 0;           // s_memrealtime ticks at 100 MHz: 200 ticks = 2 us
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(acc<0>, dim3(heads * nkb), dim3(512), LDS, 0, dq, turn, N, nkb, 0, xcc);
      if (mode == 1) hipLaunchKernelGGL(acc<1>, dim3(heads * nkb), dim3(512), LDS, 0, dq, turn, N, nkb, 0, xcc);
      if (mode == 2) hipLaunchKernelGGL(acc<2>, dim3(heads * nkb), dim3(512), LDS, 0, dq, turn, N, nkb, 0, xcc);
      if (mode == 3) hipLaunchKernelGGL(acc<3>, dim3(heads * nkb), dim3(512), LDS, 0, dq, turn, N, nkb, 0, xcc);
      if (mode >= 4 && mode < 7) hipLaunchKernelGGL(acc<4>, dim3(heads * nkb), dim3(512), LDS, 0, dq, turn, N, nkb, cyc, xcc);
      if (mode >= 7) hipLaunchKernelGGL(acc<5>, dim3(heads * nkb), dim3(512), LDS, 0, dq, turn, N, nkb, cyc, xcc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // every element of query tile qt was added (qt / per_kb + 1) times: check the ordered and atomic forms exactly
    hipMemcpy(h, dq, bytes, hipMemcpyDeviceToHost);
    long long wrong = 0;
    for (int hd = 0; hd < heads; hd += 17)
      for (int qt = 0; qt < ntile; ++qt) {
        const float want = (float)(qt / (ntile / nkb) + 1);
        for (int e = 0; e < 64 * 128; e += 97) {
          const float got = h[(size_t)hd * N * 128 + (size_t)qt * 64 * 128 + e];
          if (got != want && mode == 1 && wrong < 12 * 1000 && (wrong % 1000) == 0) printf("      head %d tile %d elem %d: %g, expected %g\n", hd, qt, e, got, want);
          wrong += got != want;
        }
      }
    double per_tile_us = 0;
    if (mode >= 4) per_tile_us = (mode >= 7 ? mode - 6 : mode - 3) * 2.0;
    printf("%-48s %8.3f ms for %.2f GB added = %7.1f GB/s   (sampled sums wrong: %lld%s)", names[mode], best, moved / 1e9, moved / best / 1e6, wrong,
           mode == 2 ? ", expected: racy" : "");
    if (mode >= 4) printf("   [the work alone: %.3f ms = %.0f us x %lld pairs / 256 CUs]", per_tile_us * heads * pairs / 256 / 1e3, per_tile_us, heads * pairs);
    printf("\n");
    if (mode == 1) {                                     // does block id & 7 name the XCD?  (the placement the same-XCD protocol relies on)
      unsigned* hx = (unsigned*)malloc(heads * nkb * 4);
      hipMemcpy(hx, xcc, heads * nkb * 4, hipMemcpyDeviceToHost);
      int off = 0, hist[16] = {0};
      for (int i = 0; i < heads * nkb; ++i) { off += hx[i] != hx[i & 7]; hist[hx[i] & 15]++; }
      printf("    workgroups whose XCC_ID differs from that of workgroup (id & 7): %d of %d;  workgroups per XCC_ID:", off, heads * nkb);
      for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
      printf("\n");
    }
  }
  return 0;
}
