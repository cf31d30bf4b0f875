// tfa_merge.hip — split-KV merge (include/tfa.h: tfa_merge).  HBM-bound elementwise kernel: every thread owns 4
// consecutive d of one row, reads the row's nparts LSEs (broadcast within the row's threads) and nparts x 16 bytes of O.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "tfa.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename OUT>
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ o_parts, const float* __restrict__ lse_parts, int nparts,
                                                    long long rows, int D, long long ostride, long long lstride, OUT* __restrict__ out,
                                                    float* __restrict__ lse_out) {
  const int tpr = D / 4;                                  // threads per row
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = gid / tpr;
  const int c = (int)(gid - row * tpr);
  if (row >= rows) return;
  float m = -INFINITY;
  for (int p = 0; p < nparts; ++p) {
    const float l = lse_parts[p * lstride + row];
    if (l != INFINITY) m = fmaxf(m, l);                   // +inf marks an empty part (tfa_fwd's convention for empty rows)
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float wsum = 0.f;
  if (m != -INFINITY) {
    // branch-free body (an empty part gets weight 0; tfa_fwd writes O = 0 for it) so that the loads of several parts
    // are in flight together: the kernel is a pure HBM stream
#pragma unroll 8
    for (int p = 0; p < nparts; ++p) {
      const float l = lse_parts[p * lstride + row];
      const f32x4 o = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(o_parts + p * ostride + row * D + c * 4));
      const float w = (l == INFINITY) ? 0.f : __expf(l - m);
      acc += w * o;
      wsum += w;
    }
  }
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) out[row * D + c * 4 + e] = (OUT)(acc[e] * inv);
  if (lse_out && c == 0) lse_out[row] = wsum > 0.f ? m + __logf(wsum) : INFINITY;
}

}  // namespace

extern "C" int tfa_merge(const float* o_parts, const float* lse_parts, int nparts, int64_t rows, int D, int64_t o_part_stride,
                         int64_t lse_part_stride, void* out, int out_dtype, float* lse_out, void* stream) {
  if (!o_parts || !lse_parts || !out) return TFA_ERR_NULL;
  if (nparts <= 0 || rows <= 0) return TFA_ERR_SHAPE;
  if (D < 4 || D > 256 || (D % 4) != 0) return TFA_ERR_HEAD_DIM;   // a thread owns one 16-byte chunk of fp32
  if (o_part_stride < rows * D || lse_part_stride < rows) return TFA_ERR_STRIDE;
  if (((uintptr_t)o_parts | (uintptr_t)out) & 15) return TFA_ERR_ALIGN;
  if ((o_part_stride * 4) % 16 != 0) return TFA_ERR_STRIDE;
  if (out_dtype != TFA_F16 && out_dtype != TFA_BF16 && out_dtype != TFA_F32) return TFA_ERR_DTYPE;
  const long long threads = rows * (D / 4);
  if (threads / 256 >= (long long)0x7fffffff) return TFA_ERR_SHAPE;
  const int grid = (int)((threads + 255) / 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();   // report THIS launch's status, not a stale sticky error
  if (out_dtype == TFA_BF16)
    hipLaunchKernelGGL(merge_kernel<__bf16>, dim3(grid), dim3(256), 0, s, o_parts, lse_parts, nparts, (long long)rows, D, (long long)o_part_stride,
                       (long long)lse_part_stride, reinterpret_cast<__bf16*>(out), lse_out);
  else if (out_dtype == TFA_F16)
    hipLaunchKernelGGL(merge_kernel<_Float16>, dim3(grid), dim3(256), 0, s, o_parts, lse_parts, nparts, (long long)rows, D, (long long)o_part_stride,
                       (long long)lse_part_stride, reinterpret_cast<_Float16*>(out), lse_out);
  else
    hipLaunchKernelGGL(merge_kernel<float>, dim3(grid), dim3(256), 0, s, o_parts, lse_parts, nparts, (long long)rows, D, (long long)o_part_stride,
                       (long long)lse_part_stride, reinterpret_cast<float*>(out), lse_out);
  return (int)hipGetLastError();
}
