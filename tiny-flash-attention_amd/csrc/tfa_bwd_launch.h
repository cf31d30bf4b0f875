// tfa_bwd_launch.h — host-side declarations shared by the backward instantiation units and tfa_bwd_api.hip
#pragma once
#include <hip/hip_runtime.h>
#include "tfa_bwd_kernel.h"
#include "tfa_bwd_kv_kernel.h"
#include "tfa_bwd_dq_kernel.h"
#include "tfa_host_util.h"

// key groups (32 resident keys each) per workgroup of the fused dK/dV kernel: 4 (eight waves, two per SIMD, 186 registers, no
// scratch — the default) or 6 (twelve waves, three per SIMD at 168 registers with 15-18 of them spilled: 1-3 % faster in one
// measurement, 15 % SLOWER in the next build of the same kernel code once other scratch-using kernels shared the library —
// profiles/r03_bwd_kg_ab.txt; a kernel that touches scratch is at the mercy of the runtime's scratch sizing, so it stays an arm:
// -DTFA_BWD_KV_KG=6).  The workspace form is blocked by 128 keys and always uses 4.
#ifndef TFA_BWD_KV_KG
#define TFA_BWD_KV_KG 4
#endif
#define TFA_BWD_KV_KG_OF(WS) ((WS) ? 4 : TFA_BWD_KV_KG)

namespace tfa {
template <typename T, int D>
hipError_t launch_bwd(const BArgs& a, int mode, int grid, bool causal, bool f32out, hipStream_t stream, bool dry);
// the 256-wide single-gradient kernels by the number of 32-column blocks that can hold valid head-dim columns (5..8)
template <typename T, int DVB>
hipError_t launch_bwd_wide(const BArgs& a, int mode, int grid, bool causal, bool f32out, hipStream_t stream, bool dry);
// dK and dV in one launch (tfa_bwd_kv_kernel.h): grid = B * Hk * ceil(Nk / 128)
template <typename T, int D>
hipError_t launch_bwd_kv(const BArgs& a, int grid, bool causal, bool f32out, hipStream_t stream, bool dry);   // a.ws != nullptr: also writes dS
// dQ = scale * dS . K from the workspace (tfa_bwd_dq_kernel.h): grid = B * H * ceil(Nq / 256)
template <typename T, int D>
hipError_t launch_bwd_dq_ws(const BArgs& a, int grid, bool causal, bool f32out, hipStream_t stream, bool dry);
template <typename T, int D>
hipError_t launch_delta(const void* o, const void* dout, float* delta, const long long* os, const long long* ds, int H, int Nq, long long rows,
                        int dv, hipStream_t stream, bool dry);
}  // namespace tfa
