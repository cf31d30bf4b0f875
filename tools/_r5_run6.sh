#!/bin/bash
# GPU call 6 (round 5): the hand-scheduled steady-state loop of il8 — bits against the compiler-scheduled build, speed (random and zeros), the parity suite
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARGS="hipcc=$L/lib_pre_asm/libtfa_hip.so:30 asm=$L/lib/libtfa_hip.so:30"
( echo "== random data"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4,cfg5,n2k,n1k --rounds 7 --iters 30 --check
  echo "== zeros"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4 --rounds 5 --iters 30 --data zeros --check ) > gpurun_out/r05_asm_loop_ab.txt 2>&1
cat gpurun_out/r05_asm_loop_ab.txt | grep -v amdgpu.ids
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r05_gpu_tests_asm.log 2>&1
tail -4 gpurun_out/r05_gpu_tests_asm.log
