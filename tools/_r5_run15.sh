#!/bin/bash
# GPU call 15 (round 5): is the decode path slower than at the round's start (bench secondary: 0.81 vs 0.89 of HBM)?  A/B of the libraries; the in-library MFMA probe without torch
cd /root/repo; mkdir -p gpurun_out; L=tiny-flash-attention_amd
( timeout 300 python tools/ab_multi.py pre_asm=$L/lib_pre_asm/libtfa_hip.so:-1 pre_x4=$L/lib_pre_x4/libtfa_hip.so:-1 now=$L/lib/libtfa_hip.so:-1 --cfgs decode,decmha --rounds 7 --iters 50 --check ) > gpurun_out/r05_decode_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_decode_ab.txt
( echo "== tools/probe_ceiling_lib (no torch)"; timeout 60 tools/probe_ceiling_lib; echo "== tools/probe_mfma_power 2.0"; timeout 120 tools/probe_mfma_power 2.0 | grep -E "sustained" | head -8 ) > gpurun_out/r05_mfma_ceiling_crosscheck3.txt 2>&1
cat gpurun_out/r05_mfma_ceiling_crosscheck3.txt
