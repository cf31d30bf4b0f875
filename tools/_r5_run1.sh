#!/bin/bash
# GPU call 1 (round 5): slope of the il8 tile body's time against extra do-nothing instructions per class
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARGS="base=$L/lib/libtfa_hip.so:30 nop32=$L/lib_pad0/libtfa_hip.so:30 nop64=$L/lib_pad0x2/libtfa_hip.so:30 salu32=$L/lib_pad1/libtfa_hip.so:30 valu32=$L/lib_pad2/libtfa_hip.so:30 wait32=$L/lib_pad3/libtfa_hip.so:30"
( rocm-smi --showpower --showclocks | head -30
  echo "== random data"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4 --rounds 5 --iters 30
  echo "== zeros"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc --rounds 5 --iters 30 --data zeros ) > gpurun_out/r5_pad_slope.txt 2>&1
tail -30 gpurun_out/r5_pad_slope.txt
