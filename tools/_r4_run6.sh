#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 300 python tools/r4_lone.py --variants 45,46,47 --hk 8 --rows 256,128 > gpurun_out/r4/lone3.txt 2>&1
timeout 400 python tools/ab_variants.py --variants 30,55,56,57,58,59,60 --cfgs cfg3nc,cfg4 --rounds 5 --iters 40 --data zeros > gpurun_out/r4/ab_ablate.txt 2>&1
timeout 400 python tools/ab_variants.py --variants 30,55,56,57,58,59,60 --cfgs cfg3nc,cfg4 --rounds 5 --iters 40 >> gpurun_out/r4/ab_ablate.txt 2>&1
cat gpurun_out/r4/lone3.txt gpurun_out/r4/ab_ablate.txt
