#!/bin/bash
# usage: tools/build_x4_variant.sh NAME "-DTFA_X4_...=..." : lib_NAME/libtfa_hip.so = product objects + the bf16/D=128 x4 unit
# rebuilt with the given flags (for tools/ab_multi.py).
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p ../build_$NAME ../lib_$NAME
make EXTRA="$FLAGS" OBJDIR=../build_$NAME OUTDIR=../lib_$NAME ../build_$NAME/tfa_x4_inst_bf16_128.o 2>&1 | grep -E "error|FAILED" || true
[ -f ../build_$NAME/tfa_x4_inst_bf16_128.o ] || { echo "build of $NAME failed"; exit 1; }
objs=$(ls ../build/*.o | grep -v tfa_x4_inst_bf16_128)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build_$NAME/tfa_x4_inst_bf16_128.o -o ../lib_$NAME/libtfa_hip.so
rm -rf ../build_$NAME/x4_bf16_128
echo "built lib_$NAME"
