#!/bin/bash
# Round-5 arm libraries: tools/r5_arm.sh <name> [extra hipcc flags...]  ->  tiny-flash-attention_amd/lib_<name>/libtfa_hip.so
# = the product objects of build/ with the bf16 D=128 forward units rebuilt with the extra flags (e.g. -DTFA_IL_PAD=1 -DTFA_IL_PADKIND=0).
# The product library in lib/ is not touched.  A/B: python tools/ab_multi.py base=.../lib/libtfa_hip.so:30 arm=.../lib_<name>/libtfa_hip.so:30 --check
set -e
name=$1; shift
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
mkdir -p ../build_$name ../lib_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -Wno-inline-asm -Wno-unused-variable -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true $*"
UNITS="${R5_UNITS:-tfa_fwd_inst_bf16_128_c1 tfa_fwd_inst_bf16_128_c0}"
for u in $UNITS; do
  ( /opt/rocm/bin/hipcc $FLAGS -c $u.hip -o ../build_$name/$u.o > ../build_$name/$u.log 2>&1 || echo "FAILED $u" ) &
done
wait
skip=""
for u in $UNITS; do [ -f ../build_$name/$u.o ] || { echo "BUILD FAILED: $u"; tail -5 ../build_$name/$u.log; exit 1; }; skip="$skip -e /$u.o"; done
objs=$(ls ../build/*.o | grep -v $skip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build_$name/*.o -o ../lib_$name/libtfa_hip.so
echo "built lib_$name/libtfa_hip.so"
