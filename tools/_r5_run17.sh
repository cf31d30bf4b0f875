#!/bin/bash
# GPU call 17 (round 5): VALU prelude behind the tile barrier (4 / 8 / 12 elements scaled and exponentiated in front of the first MFMA)
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARGS="base=$L/lib/libtfa_hip.so:30 pre4=$L/lib_pre4/libtfa_hip.so:30 pre8=$L/lib_pre8/libtfa_hip.so:30 pre12=$L/lib_pre12/libtfa_hip.so:30"
( echo "== random"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4 --rounds 7 --iters 30 --check
  echo "== zeros"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc --rounds 5 --iters 30 --data zeros ) > gpurun_out/r05_asm_prelude_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_prelude_ab.txt
