// tools/probe_oob.hip — what does an LDS-DMA (buffer_load_dwordx4 ... lds) do for lanes whose offset is outside the buffer
// descriptor's range: write zeros to LDS, or leave the LDS bytes alone?  (The attention kernels rely on the answer for the
// rows of a K/V tile beyond the end of the sequence.)   build: hipcc --offload-arch=gfx950 -O2 -o probe_oob probe_oob.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out, int mode) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x;
  reinterpret_cast<u32x4*>(smem)[lane] = u32x4{0x7f807f80u, 0x7f807f80u, 0x7f807f80u, 0x7f807f80u};   // a NaN pattern in every 16-bit half
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  int voff = lane * 16;                         // lanes >= nbytes/16 are out of range
  if (mode == 1 && lane >= 32) voff = (int)0x80000000u;   // the kernels' explicit "out of range" offset
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
  __syncthreads();
  u32x4 v = reinterpret_cast<u32x4*>(smem)[lane];
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  unsigned *src, *out, h[256];
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  unsigned hs[1024]; for (int i = 0; i < 1024; ++i) hs[i] = 0x11110000u + i;
  hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, src, 512u, out, mode);   // 512 bytes in range = lanes 0..31
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    printf("mode %d (%s): lane 0 -> %08x, lane 31 -> %08x, lane 32 -> %08x %08x, lane 63 -> %08x  => out-of-range lanes %s\n", mode,
           mode ? "offset 0x80000000" : "offset beyond num_records", h[0], h[31 * 4], h[32 * 4], h[32 * 4 + 1], h[63 * 4],
           h[32 * 4] == 0 ? "WRITE ZEROS" : (h[32 * 4] == 0x7f807f80u ? "LEAVE THE LDS BYTES ALONE" : "write something else"));
  }
  return 0;
}
