#!/bin/bash
# GPU call 10 (round 5): the fp16 twin of the hand-scheduled loop: bits + speed against the compiler-scheduled build; exact flag on fp16; parity suite
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
( timeout 300 python tools/ab_multi.py hipcc=$L/lib_pre_asm/libtfa_hip.so:30 asm=$L/lib/libtfa_hip.so:30 --cfgs f16c,f16nc,cfg3 --rounds 5 --iters 30 --check
  timeout 300 python tools/ab_multi.py hipcc38=$L/lib_pre_asm/libtfa_hip.so:38 asm38=$L/lib/libtfa_hip.so:38 --cfgs f16c,f16nc --rounds 5 --iters 30 --check ) > gpurun_out/r05_asm_f16_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_f16_ab.txt
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_gpu_tests_asm3.log 2>&1
tail -3 gpurun_out/r05_gpu_tests_asm3.log
