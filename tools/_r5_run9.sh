#!/bin/bash
# GPU call 9 (round 5): fuzz on the final kernels, PMC passes (default + exact), the driver's bench command twice, rocprofv3 kernel stats (csv)
cd /root/repo; mkdir -p gpurun_out
( timeout 500 python tools/fuzz_fwd.py --big --n 300 --seed 5200 2>&1 | tail -3; timeout 300 python tools/fuzz_fwd.py --n 400 --seed 5300 2>&1 | tail -3 ) > gpurun_out/r05_fuzz2.txt 2>&1
B="python /root/repo/bench.py --config cfg3 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --precondition-s 0.5"
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_default_cfg3 -- $B ) > gpurun_out/r05_pmc_default.log 2>&1
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_exact_cfg3 -- $B --variant 38 ) > gpurun_out/r05_pmc_exact.log 2>&1
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_default_cfg4 -- python /root/repo/bench.py --config cfg4 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --precondition-s 0.5 ) > gpurun_out/r05_pmc_default4.log 2>&1
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_a.json 2> gpurun_out/r05_bench_driver_a.err
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r05_bench_driver_b.json 2> gpurun_out/r05_bench_driver_b.err
cd /tmp && export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5_prof2 -o r5 -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline ) > /root/repo/gpurun_out/r5_prof2.log 2>&1
cd /root/repo; cat gpurun_out/r05_fuzz2.txt; grep -E "busy|per_mfma" gpurun_out/r05_pmc_default_cfg3.txt gpurun_out/r05_pmc_exact_cfg3.txt; python3 -c "
import json
for f in ('a','b'):
    j=json.load(open('gpurun_out/r05_bench_driver_%s.json'%f)); r=j['roofline']; print(f, round(j['value'],1), round(r['frac'],4), r['mfma_only_ceiling_random_data'], r['frac_of_mfma_only_ceiling'], j['box'])
"; find gpurun_out/r5_prof2 -name "*kernel_stats.csv" | head -2
