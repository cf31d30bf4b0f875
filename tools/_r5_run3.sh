#!/bin/bash
# GPU call 3 (round 5): exact-il8 — how many softmax elements the re-basing body keeps in its QK^T half (NE1X), packed O multiply arm
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
ARGS="ne21=$L/lib/libtfa_hip.so:38 ne10=$L/lib_ne10/libtfa_hip.so:38 ne12=$L/lib_ne12/libtfa_hip.so:38 ne14=$L/lib_ne14/libtfa_hip.so:38 ne10pk=$L/lib_ne10pk/libtfa_hip.so:38 lazy=$L/lib/libtfa_hip.so:30"
( echo "== random data"; timeout 300 python tools/ab_multi.py $ARGS --cfgs cfg3,cfg3nc,cfg4 --rounds 5 --iters 30 --check
  ) > gpurun_out/r5_exact_il8_ne1x.txt 2>&1
tail -30 gpurun_out/r5_exact_il8_ne1x.txt
