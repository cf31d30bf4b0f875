#!/bin/bash
# Round-4 arms library (AFTER `git apply experiments/r04_il8_arms.patch`): lib_x/libtfa_hip.so = the product objects of build/ with the bf16 D=128 forward units, tfa_api and the ablation unit rebuilt
# with -DTFA_R4_ARMS (the A/B arms 38.. of tfa_launch.h) — about a minute, the product library in lib/ is not touched.
#   tools/r4_quick.sh            then   TFA_LIB=$PWD/tiny-flash-attention_amd/lib_x/libtfa_hip.so python tools/ab_variants.py --variants 30,61 ...
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
mkdir -p ../build_x ../lib_x
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -I../../experiments/csrc -Wno-unused-function -Wno-inline-asm -Wno-unused-variable -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true -DTFA_R4_ARMS"
for u in tfa_fwd_inst_bf16_128_c1 tfa_fwd_inst_bf16_128_c0 tfa_api; do
  ( /opt/rocm/bin/hipcc $FLAGS -c $u.hip -o ../build_x/$u.o > /tmp/r4x_$u.log 2>&1 || echo "FAILED $u" ) &
done
( /opt/rocm/bin/hipcc $FLAGS -c ../../experiments/csrc/tfa_ilab_inst_bf16_128.hip -o ../build_x/tfa_ilab_inst_bf16_128.o > /tmp/r4x_ilab.log 2>&1 || echo "FAILED ilab" ) &
wait
for u in tfa_fwd_inst_bf16_128_c1 tfa_fwd_inst_bf16_128_c0 tfa_api tfa_ilab_inst_bf16_128; do [ -f ../build_x/$u.o ] || { echo "BUILD FAILED: $u"; tail -5 /tmp/r4x_*.log; exit 1; }; done
objs=$(ls ../build/*.o | grep -v "tfa_fwd_inst_bf16_128_c[01].o\|tfa_api.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build_x/*.o -o ../lib_x/libtfa_hip.so
echo "built lib_x/libtfa_hip.so"
