#!/bin/bash
# GPU call 11 (round 5): the hand-scheduled loop of the 256-wide kernel (x4-d256, variant 34): bits + speed against the compiler-scheduled build, head-dim tests
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
( echo "== random"; timeout 400 python tools/ab_multi.py hipcc=$L/lib_pre_x4/libtfa_hip.so:34 asm=$L/lib/libtfa_hip.so:34 --cfgs d256c,d256nc,d256f16c,d256n16k --rounds 5 --iters 30 --check
  echo "== zeros"; timeout 300 python tools/ab_multi.py hipcc=$L/lib_pre_x4/libtfa_hip.so:34 asm=$L/lib/libtfa_hip.so:34 --cfgs d256c,d256nc --rounds 5 --iters 30 --data zeros --check ) > gpurun_out/r05_asm_x4_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_x4_ab.txt
( timeout 1200 python -m pytest tests/ -x -q -m gpu -k "head_dim or 256 or d256 or x4 or splitkv or fuzz" 2>&1 | tail -5 ) > gpurun_out/r05_gpu_tests_x4.log 2>&1
tail -3 gpurun_out/r05_gpu_tests_x4.log
