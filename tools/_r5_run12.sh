#!/bin/bash
# GPU call 12 (round 5): LDS-DMA with scalar tile offsets (both loops) and quad fragment groups (x4 loop): bits + speed; full GPU suite
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
( echo "== x4 (variant 34): hipcc body | asm v1 (pairs, VALU offsets: 378) is not in a library any more | asm v2 (quads, scalar offsets: 346)"
  timeout 400 python tools/ab_multi.py hipcc=$L/lib_pre_x4/libtfa_hip.so:34 asm2=$L/lib/libtfa_hip.so:34 --cfgs d256c,d256nc,d256f16c,d256n16k --rounds 5 --iters 30 --check
  echo "== il8 (variant 30): asm v1 (246, VALU offsets) | asm v2 (242, scalar offsets)"
  timeout 400 python tools/ab_multi.py asm1=$L/lib_pre_x4/libtfa_hip.so:30 asm2=$L/lib/libtfa_hip.so:30 --cfgs cfg3,cfg3nc,cfg4 --rounds 7 --iters 30 --check
  echo "== zeros"; timeout 300 python tools/ab_multi.py asm1=$L/lib_pre_x4/libtfa_hip.so:30 asm2=$L/lib/libtfa_hip.so:30 --cfgs cfg3,cfg3nc --rounds 5 --iters 30 --data zeros
  timeout 300 python tools/ab_multi.py hipcc=$L/lib_pre_x4/libtfa_hip.so:34 asm2=$L/lib/libtfa_hip.so:34 --cfgs d256c,d256nc --rounds 5 --iters 30 --data zeros ) > gpurun_out/r05_asm_v2_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_v2_ab.txt
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_gpu_tests_asm4.log 2>&1
tail -3 gpurun_out/r05_gpu_tests_asm4.log
