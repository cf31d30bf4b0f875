#!/bin/bash
# Compile ONE instantiation of the x4 kernel (bf16, D=128, 16-bit out; ONE_CAUSAL / ONE_AB / extra -D via $EXTRA) and report
# registers, spills and the hot-loop statistics.  Output in /tmp/x4/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/x4 && cd /tmp/x4
cat > one.hip <<'HIP'
#include "tfa_fwd_kernel_x4.h"
namespace tfa {
#ifndef ONE_CAUSAL
#define ONE_CAUSAL false
#endif
#ifndef ONE_AB
#define ONE_AB 0
#endif
#ifndef ONE_D
#define ONE_D 128
#endif
template __global__ void fwd_kernel_x4<__bf16, ONE_D, ONE_CAUSAL, false, VF_PAIR | (ONE_D <= 128 ? VF_X4_EPI : 0), ONE_AB>(const KArgs);
}
HIP
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/tiny-flash-attention_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true --cuda-device-only -save-temps -Rpass-analysis=kernel-resource-usage $EXTRA -c one.hip -o one.o 2> res.txt || { grep -E "error" -A3 res.txt | head -30; exit 1; }
grep -E "VGPRs:|AGPRs|Spill|ScratchSize" res.txt | sed 's/remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
S=$(ls one-hip-amdgcn*.s | head -1)
awk '/^_ZN3tfa13fwd_kernel_x4/,/s_endpgm/' $S > one.s
echo "one.s: $(wc -l < one.s) lines, scratch ops $(grep -c scratch_ one.s), v_accvgpr $(grep -c v_accvgpr one.s), mfma $(grep -c v_mfma one.s)"
python3 $ROOT/experiments/tools/x4_loop_stats.py /tmp/x4/one.s --min 40 | grep -v "VALU ops"
