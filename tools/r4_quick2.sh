#!/bin/bash
# Arms library for arms kept in the tree under -DTFA_R4_ARMS (no patch to apply): lib_x/libtfa_hip.so = the product objects of build/ with the
# bf16 D=128 forward units and tfa_api rebuilt with -DTFA_R4_ARMS.  About a minute; the product library in lib/ is not touched.
set -e
cd "$(dirname "$0")/../tiny-flash-attention_amd/csrc"
rm -rf ../build_x; mkdir -p ../build_x ../lib_x
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -Wno-inline-asm -Wno-unused-variable -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true -DTFA_R4_ARMS"
for u in tfa_fwd_inst_bf16_128_c1 tfa_fwd_inst_bf16_128_c0 tfa_api; do
  ( /opt/rocm/bin/hipcc $FLAGS -c $u.hip -o ../build_x/$u.o > /tmp/r4x_$u.log 2>&1 || echo "FAILED $u" ) &
done
wait
for u in tfa_fwd_inst_bf16_128_c1 tfa_fwd_inst_bf16_128_c0 tfa_api; do [ -f ../build_x/$u.o ] || { echo "BUILD FAILED: $u"; tail -5 /tmp/r4x_$u.log; exit 1; }; done
objs=$(ls ../build/*.o | grep -v "tfa_fwd_inst_bf16_128_c[01].o\|tfa_api.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../build_x/*.o -o ../lib_x/libtfa_hip.so
echo "built lib_x/libtfa_hip.so"
