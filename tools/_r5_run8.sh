#!/bin/bash
# GPU call 8 (round 5): the exact-running-max loop in hand-scheduled form (variant 38): bits and speed against the compiler-scheduled build, exact tests, full suite
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARGS="hipcc38=$L/lib_pre_asm/libtfa_hip.so:38 asm38=$L/lib/libtfa_hip.so:38 asm30=$L/lib/libtfa_hip.so:30"
( echo "== random data"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4,n2k --rounds 7 --iters 30 --check ) > gpurun_out/r05_asm_exact_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_exact_ab.txt
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_gpu_tests_asm2.log 2>&1
tail -3 gpurun_out/r05_gpu_tests_asm2.log
