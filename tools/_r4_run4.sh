#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 300 python tools/r4_lone.py --variants 45,48 --hk 8 > gpurun_out/r4/lone2.txt 2>&1
timeout 300 python tools/r4_lone.py --variants 48 --hk 8 --dbg 2048 --rows 256,128 >> gpurun_out/r4/lone2.txt 2>&1
for dbg in 0 2048; do
timeout 100 python tools/trace_iters.py --shape 1,256,128,8192,0,8 --variant 48 --wg 0 --dbg $dbg >> gpurun_out/r4/iters_lone2.txt 2>&1
done
timeout 100 python tools/trace_iters.py --shape 1,256,256,8192,0,8 --variant 48 --wg 0 >> gpurun_out/r4/iters_lone2.txt 2>&1
KERNEL_FILTER=fwd_kernel timeout 600 python tools/prof_pmc.py gpurun_out/r4/pmc_lone128 -- python tools/r4_lone.py --variants 45 --rows 128 --hk 8 > gpurun_out/r4/pmc_lone128.log 2>&1
KERNEL_FILTER=fwd_kernel timeout 600 python tools/prof_pmc.py gpurun_out/r4/pmc_lone256 -- python tools/r4_lone.py --variants 45 --rows 256 --hk 8 > gpurun_out/r4/pmc_lone256.log 2>&1
cat gpurun_out/r4/lone2.txt; cut -c1-230 gpurun_out/r4/iters_lone2.txt | grep -v "^   it  *[3-9] \|^   it  1[0-9][0-9]\|\.\.\." 
