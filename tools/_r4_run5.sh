#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 400 python tools/ab_variants.py --variants 30,50,51,52,53 --cfgs cfg3,cfg3nc,cfg4 --rounds 7 --iters 40 --check > gpurun_out/r4/ab_prio.txt 2>&1
timeout 300 python tools/ab_variants.py --variants 30,50,51,52,53 --cfgs cfg3,cfg3nc --rounds 5 --iters 40 --data zeros >> gpurun_out/r4/ab_prio.txt 2>&1
timeout 100 python tools/trace_iters.py cfg3nc --variant 54 --wg 0 > gpurun_out/r4/iters_prio.txt 2>&1
KERNEL_FILTER=fwd_kernel timeout 600 python tools/prof_pmc.py gpurun_out/r4/pmc_lone128 -- python $PWD/tools/r4_lone.py --variants 45 --rows 128 --hk 8 > gpurun_out/r4/pmc_lone128.log 2>&1
KERNEL_FILTER=fwd_kernel timeout 600 python tools/prof_pmc.py gpurun_out/r4/pmc_lone256 -- python $PWD/tools/r4_lone.py --variants 45 --rows 256 --hk 8 > gpurun_out/r4/pmc_lone256.log 2>&1
cat gpurun_out/r4/ab_prio.txt; cut -c1-230 gpurun_out/r4/iters_prio.txt | tail -16
