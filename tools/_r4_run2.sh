#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 200 python tools/trace_iters.py cfg3 --wg 0,56,24 > gpurun_out/r4/iters_cfg3.txt 2>&1
timeout 200 python tools/trace_iters.py cfg3nc --wg 0 > gpurun_out/r4/iters_cfg3nc.txt 2>&1
timeout 100 python tools/ab_variants.py --variants 30,41 --cfgs cfg3 --rounds 3 --iters 20 --check > gpurun_out/r4/ab_itr.txt 2>&1
cat gpurun_out/r4/ab_itr.txt
