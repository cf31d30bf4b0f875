// timing-only ablations of the product il8 kernel, bf16 D=128 (round 4: which instruction family costs what).  Results are WRONG by
// design (operands the MFMAs read are never written, hazards are not respected): a unit of its own because the il units' build-time
// hazard audit rightly rejects these kernels.  Reached through tfa_set_variant(55..60) only (tools/ab_variants.py).
#include "tfa_launch.h"

namespace tfa {

template <int AB, bool CAUSAL>
static hipError_t launch_il_ab(const KArgs& a, hipStream_t stream, LaunchGeom* geom, bool dry) {
  constexpr int D = 128;
  constexpr int VF = VF_PAIR | VF_IL_DMASPREAD | VF_IL_EPI;
  constexpr int lds = 4 * 64 * D * 2 + 8 * 32 * D * 2;
  auto kern = fwd_kernel_il<__bf16, D, 8, CAUSAL, false, VF, AB>;
  static std::atomic<unsigned long long> attr_mask{0};
  return launch_common(kern, attr_mask, a.nbh * a.nwork, 512, lds, a, stream, geom, dry);
}
template <int AB>
static hipError_t launch_il_ab_c(const KArgs& a, bool causal, hipStream_t s, LaunchGeom* g, bool dry) {
  return causal ? launch_il_ab<AB, true>(a, s, g, dry) : launch_il_ab<AB, false>(a, s, g, dry);
}

hipError_t launch_il_ablation(const KArgs& a, int variant, bool causal, hipStream_t s, LaunchGeom* g, bool dry) {
  switch (variant) {
    case 55: return launch_il_ab_c<ILAB_NOMAX>(a, causal, s, g, dry);
    case 56: return launch_il_ab_c<ILAB_NOKREAD | ILAB_NOVREAD>(a, causal, s, g, dry);
    case 57: return launch_il_ab_c<ILAB_NODMA>(a, causal, s, g, dry);
    case 58: return launch_il_ab_c<ILAB_NOEXP | ILAB_NOMAX>(a, causal, s, g, dry);
    case 59: return launch_il_ab_c<ILAB_NOEXP | ILAB_NOMAX | ILAB_NOKREAD | ILAB_NOVREAD>(a, causal, s, g, dry);
    case 60: return launch_il_ab_c<ILAB_NOBARRIER>(a, causal, s, g, dry);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace tfa
