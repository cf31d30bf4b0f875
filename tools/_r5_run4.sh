#!/bin/bash
# GPU call 4 (round 5): full GPU suite on the rebuilt library, the driver's bench command, rocprofv3 kernel stats of the same command
cd /root/repo; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r5_gpu_tests.log 2>&1
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_driver.json 2> gpurun_out/r5_bench_driver.err
cd /tmp && export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r5_prof -o r5 -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline ) > /root/repo/gpurun_out/r5_prof.log 2>&1
cd /root/repo
find gpurun_out/r5_prof -name "*kernel_stats*" | head; tail -3 gpurun_out/r5_gpu_tests.log; head -c 600 gpurun_out/r5_bench_driver.json; tail -3 gpurun_out/r5_bench_driver.err
