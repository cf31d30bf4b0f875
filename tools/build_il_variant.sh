#!/bin/bash
# usage: [UNITS="bf16_128_c1 f16_64_c0"] tools/build_il_variant.sh NAME "-DTFA_IL_...=..." : lib_NAME/libtfa_hip.so = product objects + the listed
# il units (default bf16_128) rebuilt with the given flags (for tools/ab_multi.py name=path:30).
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
NAME=$1; FLAGS=$2; UNITS=${UNITS:-"bf16_128_c0 bf16_128_c1"}
mkdir -p ../build_$NAME ../lib_$NAME
objs=$(ls ../build/*.o)
for u in $UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -Wno-inline-asm -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true $FLAGS -c tfa_fwd_inst_$u.hip -o ../build_$NAME/tfa_fwd_inst_$u.o 2>/dev/null &
  objs=$(echo "$objs" | grep -v "tfa_fwd_inst_$u.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build_$NAME/*.o -o ../lib_$NAME/libtfa_hip.so
echo "built lib_$NAME"
