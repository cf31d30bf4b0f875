#!/bin/bash
# GPU call 13 (round 5): the x4 loop for head dims 136..224 (5, 6, 7 valid column blocks): bits + speed against the compiler-scheduled bodies; head-dim tests
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
( timeout 400 python tools/ab_multi.py hipcc=$L/lib_pre_x4b/libtfa_hip.so:34 asm=$L/lib/libtfa_hip.so:34 --cfgs d160c,d192c,d224nc,d256c --rounds 5 --iters 30 --check ) > gpurun_out/r05_asm_x4_dvb_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_x4_dvb_ab.txt
( timeout 1200 python -m pytest tests/ -x -q -m gpu -k "head_dim or 256 or d256 or x4 or splitkv or fuzz" 2>&1 | tail -5 ) > gpurun_out/r05_gpu_tests_x4b.log 2>&1
tail -3 gpurun_out/r05_gpu_tests_x4b.log
( timeout 300 python tools/fuzz_fwd.py --n 300 --seed 5400 2>&1 | tail -2 ) > gpurun_out/r05_fuzz3.txt 2>&1; cat gpurun_out/r05_fuzz3.txt
