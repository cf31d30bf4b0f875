#!/bin/bash
# usage: tools/build_x4_variant.sh NAME "-DTFA_X4_...=..." : lib_NAME/libtfa_hip.so = product objects + the four bf16/D=128 x4 units
# (causal x output type) rebuilt with the given flags (for tools/ab_multi.py).
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p ../build_$NAME ../lib_$NAME
units=""; for c in c0 c1; do for o in o16 o32; do units="$units ../build_$NAME/tfa_x4_inst_bf16_128_${c}_${o}.o"; done; done
make -j4 EXTRA="$FLAGS" OBJDIR=../build_$NAME OUTDIR=../lib_$NAME $units 2>&1 | grep -E "error|FAILED" || true
for u in $units; do [ -f $u ] || { echo "build of $NAME failed ($u)"; exit 1; }; done
objs=$(ls ../build/*.o | grep -v tfa_x4_inst_bf16_128_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $units -o ../lib_$NAME/libtfa_hip.so
echo "built lib_$NAME"
