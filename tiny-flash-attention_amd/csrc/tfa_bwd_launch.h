// tfa_bwd_launch.h — host-side declarations shared by the backward instantiation units and tfa_bwd_api.hip
#pragma once
#include <hip/hip_runtime.h>
#include "tfa_bwd_kernel.h"
#include "tfa_bwd_kv_kernel.h"
#include "tfa_bwd_dq_kernel.h"
#include "tfa_host_util.h"

// key groups (32 resident keys each) per workgroup of the fused dK/dV kernel: 4 (eight waves, two per SIMD) or 6 (twelve waves,
// three per SIMD, 168 registers each); the workspace form is blocked by 128 keys and stays at 4
#ifndef TFA_BWD_KV_KG
#define TFA_BWD_KV_KG 6
#endif
#define TFA_BWD_KV_KG_OF(WS) ((WS) ? 4 : TFA_BWD_KV_KG)

namespace tfa {
template <typename T, int D>
hipError_t launch_bwd(const BArgs& a, int mode, int grid, bool causal, bool f32out, hipStream_t stream, bool dry);
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
