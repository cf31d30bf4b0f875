// tfa_launch.h — host-side dispatch table shared by the per-(dtype, head-dim) instantiation units.
#pragma once
#include <hip/hip_runtime.h>
#include "tfa_fwd_kernel.h"

namespace tfa {

struct Variant {
  const char* name;
  int nw;   // waves per workgroup (32 query rows each)
  int vf;   // VF_* flags
};

// Keep in sync with the switch in tfa_fwd_inst.inc
static const Variant kVariants[] = {
    {"w8-gatherV (bring-up: 16-bit LDS gathers for V, no transpose read)", 8, 0},
    {"w8-trV (8 waves x 32 rows, ds_read_b64_tr_b16 for V)", 8, VF_TRREAD},
    {"w4-trV (4 waves x 32 rows, 2 workgroups/CU)", 4, VF_TRREAD},
    {"w8-trV-alwaysrescale (no exact alpha==1 skip)", 8, VF_TRREAD | VF_NOSKIP},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr int kDefaultVariant = 1;

struct LaunchGeom {
  int grid, block, lds;
};

template <typename T, int D>
hipError_t launch_fwd(const KArgs& a, bool causal, bool f32out, int variant, hipStream_t stream, LaunchGeom* geom, bool dry);

static inline int block_m_of(int variant) { return kVariants[variant].nw * 32; }

}  // namespace tfa
