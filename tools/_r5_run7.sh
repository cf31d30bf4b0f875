#!/bin/bash
# GPU call 7 (round 5): schedule knobs of the hand-scheduled loop (tools/gen_il_asm_loop.py TFA_GEN_*): elements behind the QK^T MFMAs (16 / 21 / 26),
# LDS-DMA pieces behind MFMAs 6..9 instead of 0..3, exp2 two slots ahead instead of one
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARGS="base=$L/lib/libtfa_hip.so:30 ne16=$L/lib_ne16/libtfa_hip.so:30 ne26=$L/lib_ne26/libtfa_hip.so:30 dma6=$L/lib_dma6/libtfa_hip.so:30 exp2=$L/lib_exp2/libtfa_hip.so:30"
( echo "== random data"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4 --rounds 7 --iters 30 --check
  echo "== zeros"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc --rounds 5 --iters 30 --data zeros ) > gpurun_out/r05_asm_knobs_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_asm_knobs_ab.txt
