#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 300 python tools/r4_lone.py --variants 45,46,47 > gpurun_out/r4/lone.txt 2>&1
timeout 300 python tools/ab_variants.py --variants 30,42,43 --cfgs cfg3,cfg3nc,cfg4 --rounds 7 --iters 40 --check > gpurun_out/r4/ab_pf.txt 2>&1
timeout 200 python tools/ab_variants.py --variants 30,42,43 --cfgs cfg3,cfg3nc --rounds 5 --iters 40 --data zeros >> gpurun_out/r4/ab_pf.txt 2>&1
timeout 100 python tools/trace_iters.py --shape 1,256,128,8192,0 --variant 48 --wg 0 > gpurun_out/r4/iters_lone.txt 2>&1
timeout 100 python tools/trace_iters.py --shape 1,256,128,8192,0 --variant 49 --wg 0 >> gpurun_out/r4/iters_lone.txt 2>&1
timeout 100 python tools/trace_iters.py --shape 1,256,256,8192,0 --variant 48 --wg 0 >> gpurun_out/r4/iters_lone.txt 2>&1
KERNEL_FILTER=fwd_kernel timeout 600 python tools/prof_pmc.py gpurun_out/r4/pmc_lone128 -- python tools/r4_lone.py --variants 45 --rows 128 > gpurun_out/r4/pmc_lone128.log 2>&1
KERNEL_FILTER=fwd_kernel timeout 600 python tools/prof_pmc.py gpurun_out/r4/pmc_lone256 -- python tools/r4_lone.py --variants 45 --rows 256 > gpurun_out/r4/pmc_lone256.log 2>&1
cat gpurun_out/r4/lone.txt gpurun_out/r4/ab_pf.txt
