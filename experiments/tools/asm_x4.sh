#!/bin/bash
# Compile one forward instantiation unit with -save-temps and report registers / scratch of the x4 kernels, then extract the
# non-causal bf16 D=128 x4 kernel's assembly to build/asm/x4_nc.s (tools/x4_loop_stats.py reads it).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tiny-flash-attention_amd/build/asm && cd $ROOT/tiny-flash-attention_amd/build/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/tiny-flash-attention_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true -save-temps -Rpass-analysis=kernel-resource-usage $EXTRA -c $ROOT/tiny-flash-attention_amd/csrc/tfa_fwd_inst_bf16_128_c0.hip -o /dev/null 2> res_bf16_128.txt || { grep error res_bf16_128.txt | head; exit 1; }
grep -A12 "fwd_kernel_x4" res_bf16_128.txt | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize" | sed 's/remark: [^ ]* *//; s/\[-Rpass.*//'
S=$(ls *gfx950.s | head -1)
awk '/^_ZN3tfa13fwd_kernel_x4IDF16bLi128ELb0ELb0ELi[0-9]*ELi0EEEvNS_5KArgsE:/,/s_endpgm/' $S > x4_nc.s
awk '/^_ZN3tfa13fwd_kernel_x4IDF16bLi128ELb1ELb0ELi[0-9]*ELi0EEEvNS_5KArgsE:/,/s_endpgm/' $S > x4_c.s
echo "x4_nc.s: $(wc -l < x4_nc.s) lines, scratch ops $(grep -c scratch_ x4_nc.s), v_accvgpr $(grep -c v_accvgpr x4_nc.s), mfma $(grep -c v_mfma x4_nc.s), branches $(grep -c s_cbranch x4_nc.s)"
