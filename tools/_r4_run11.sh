#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
export TFA_LIB=$PWD/tiny-flash-attention_amd/lib_x/libtfa_hip.so
timeout 300 python tools/r4_bits.py --base 39 --arms 30,63 --dtypes bf16 > gpurun_out/r4/bits11.txt 2>&1; tail -3 gpurun_out/r4/bits11.txt
timeout 400 python tools/ab_variants.py --variants 39,30,63 --cfgs cfg3,n2k,cfg5,cfg4c --rounds 7 --iters 40 --check > gpurun_out/r4/ab_pref3.txt 2>&1
timeout 300 python tools/ab_variants.py --variants 39,30,63 --cfgs cfg3 --rounds 5 --iters 40 --data zeros >> gpurun_out/r4/ab_pref3.txt 2>&1
cat gpurun_out/r4/ab_pref3.txt
for v in 30 63; do timeout 120 python tools/trace_passes.py $v 2>&1 | grep -v amdgpu.ids | head -3;  timeout 120 python tools/trace_passes.py $v pass1 2>&1 | grep -v amdgpu.ids | head -3; done
