#!/bin/bash
# usage: tools/build_il_variant.sh NAME "-DTFA_IL_...=..." : lib_NAME/libtfa_hip.so = product objects + the bf16/D=128 il unit
# rebuilt with the given flags (for tools/ab_multi.py name=path:30).
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p ../build_$NAME ../lib_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -Wno-inline-asm -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true $FLAGS -c tfa_fwd_inst_bf16_128.hip -o ../build_$NAME/tfa_fwd_inst_bf16_128.o 2>/dev/null
objs=$(ls ../build/*.o | grep -v tfa_fwd_inst_bf16_128)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build_$NAME/tfa_fwd_inst_bf16_128.o -o ../lib_$NAME/libtfa_hip.so
echo "built lib_$NAME"
