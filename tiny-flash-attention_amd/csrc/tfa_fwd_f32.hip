// tfa_fwd_f32.hip — the forward pass on fp32 tensors: the correctness path behind the reference's fp32 fixtures.
//
// The reference's CPU sibling (`_kernels.flash_attn` / `naive_attn`, flash_attention_c/csrc/attn.cpp:35-169,237-262) and the fp32 arm of its toy
// CUDA entry (flash_attention_cuda/csrc/flash_attention.cu:411 AT_DISPATCH_FLOATING_TYPES_AND_HALF) take fp32 tensors; their own test script
// (flash_attention_c/test.py:35-48) feeds torch.rand fp32.  This kernel serves those calls on the GPU with fp32 arithmetic end to end
// (v_mfma_f32_32x32x2_f32: exact f32 products and accumulation, 1/16 of the bf16 MFMA rate — a correctness path, not a speed path), so that the
// scripts run with only `device="cuda"` changed and meet the reference's own fp32 results to ~1e-6.
//
// One wave per 32 query rows, no LDS, no synchronisation; both GEMMs in the swapped form of the 16-bit kernels (tfa_fwd_kernel.h):
//   S^T[key][row] = sum_d K[key][d] Q[row][d]       A = K (lane: key = lane & 31, one d of the pair), B = Q^T held in registers
//   O^T[d][row]  += sum_key V[key][d] P[row][key]   B = P straight from the S^T accumulator (the lane's own 16 keys), A = V loaded to match
// with the exact running maximum of the reference's loop (attn.cpp:138-160).  Strided (B,H,N,D) / (B,N,H,D), GQA, Nq != Nk with the bottom-right
// causal mask (attn.cpp:121-124), split-KV arguments (kv_offset / nk_total) as in tfa_fwd; head dims: multiples of 4 up to 256.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <atomic>

#include "tfa.h"
#include "tfa_fwd_kernel.h"
#include "tfa_host_util.h"

namespace tfa {

// DM: compiled width (64 / 128 / 256); p.dv = the valid head dim (columns beyond it read as zeros and are never stored)
template <int DM, bool CAUSAL>
__global__ __launch_bounds__(256) void fwd_kernel_f32(const KArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = lane & 31, hi = lane >> 5;
  const int nblk = (p.Nq + 31) / 32;
  const long long gid = (long long)blockIdx.x * 4 + wave;
  if (gid >= (long long)p.nbh * nblk) return;
  const int bh = (int)(gid / nblk), qb = (int)(gid % nblk);
  const int b = bh / p.H, h = bh - b * p.H, hk = h / (p.H / p.Hk);
  const int row = qb * 32 + qi;
  const int rowc = row < p.Nq ? row : p.Nq - 1;
  const float* qrow = reinterpret_cast<const float*>(p.q) + b * p.qs_b + h * p.qs_h + (long long)rowc * p.qs_n;
  const float* kb = reinterpret_cast<const float*>(p.k) + b * p.ks_b + hk * p.ks_h;
  const float* vb = reinterpret_cast<const float*>(p.v) + b * p.vs_b + hk * p.vs_h;
  constexpr int DH = DM / 2;                     // this half-wave's d range: [hi * DH, hi * DH + DH) — the MFMA's k pair is (s, DH + s)
  float qv[DH];
#pragma unroll
  for (int s4 = 0; s4 < DH; s4 += 4) {
    const int d = hi * DH + s4;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (d < p.dv) t = *reinterpret_cast<const f32x4*>(qrow + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) qv[s4 + e] = t[e];
  }
  f32x16 O[DM / 32];
#pragma unroll
  for (int d = 0; d < DM / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int shift = p.shift;
  int kv_end = p.Nk;
  constexpr bool causal = CAUSAL;
  if (causal) {
    const int lim = qb * 32 + 32 + shift;        // keys 0 .. lim-1 are visible to the block's last row
    kv_end = lim < kv_end ? lim : kv_end;
  }
  for (int k0 = 0; k0 < kv_end; k0 += 32) {
    const int key = k0 + qi, keyc = key < p.Nk ? key : p.Nk - 1;
    const float* krow = kb + (long long)keyc * p.ks_n;
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < DH; s4 += 4) {
      const int d = hi * DH + s4;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (d < p.dv) a = *reinterpret_cast<const f32x4*>(krow + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) S = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qv[s4 + e], S, 0, 0, 0);
    }
    // S[r] = S^T[key k0 + 8*(r>>2) + 4*hi + (r&3)][row]
    float x[16], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = k0 + 8 * (r >> 2) + 4 * hi + (r & 3);
      const bool dead = kk >= p.Nk || (causal && kk > row + shift);
      x[r] = dead ? -INFINITY : S[r] * p.scale;
      mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = (mn == -INFINITY) ? 1.f : expf(m - mn);       // (m = -inf, mn finite: exp(-inf) = 0)
    float ps[16], sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ps[r] = (mn == -INFINITY) ? 0.f : expf(x[r] - mn);
      sum += ps[r];
    }
    sum += __shfl_xor(sum, 32);
    l = l * alpha + sum;
    m = mn;
#pragma unroll
    for (int d = 0; d < DM / 32; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = k0 + 8 * (r >> 2) + 4 * hi + (r & 3);            // the key whose P this lane holds in ps[r]
      const bool kv_ok = kv < p.Nk;                                      // (a padded key contributes 0, not 0 * V[Nk-1]: no Inf / NaN leaks out of the clamped row)
      const float* vrow = vb + (long long)(kv_ok ? kv : p.Nk - 1) * p.vs_n;
#pragma unroll
      for (int d = 0; d < DM / 32; ++d) {
        const float a = (kv_ok && d * 32 + qi < p.dv) ? vrow[d * 32 + qi] : 0.f;
        O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ps[r], O[d], 0, 0, 0);
      }
    }
  }
  if (row >= p.Nq) return;
  const bool empty = !(l > 0.f);
  const float inv = empty ? 0.f : 1.f / l;
  if (p.lse != nullptr && hi == 0) p.lse[(long long)bh * p.Nq + row] = empty ? INFINITY : m + logf(l);
  float* orow = reinterpret_cast<float*>(p.o) + b * p.os_b + h * p.os_h + (long long)row * p.os_n;
#pragma unroll
  for (int d = 0; d < DM / 32; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = d * 32 + 8 * g + 4 * hi;
      if (c < p.dv) {
        const f32x4 o4 = {O[d][4 * g] * inv, O[d][4 * g + 1] * inv, O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + c) = o4;
      }
    }
}

// host side: called by tfa_api.hip's run() for tfa_fwd_params::dtype == TFA_F32 (validated there)
hipError_t launch_f32(const KArgs& a, bool causal, hipStream_t stream, int* grid_out, bool dry) {
  const long long waves = (long long)a.nbh * ((a.Nq + 31) / 32);
  const long long grid = (waves + 3) / 4;
  if (grid >= 0x7fffffffll) return hipErrorInvalidValue;
  if (grid_out) *grid_out = (int)grid;
  if (dry) return hipSuccess;
  (void)hipGetLastError();
#define TFA_F32_LAUNCH(DM)                                                                                              \
  do {                                                                                                                  \
    if (causal) hipLaunchKernelGGL((fwd_kernel_f32<DM, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);          \
    else hipLaunchKernelGGL((fwd_kernel_f32<DM, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);                \
  } while (0)
  if (a.dv <= 64) TFA_F32_LAUNCH(64);
  else if (a.dv <= 128) TFA_F32_LAUNCH(128);
  else TFA_F32_LAUNCH(256);
#undef TFA_F32_LAUNCH
  return hipGetLastError();
}

}  // namespace tfa
