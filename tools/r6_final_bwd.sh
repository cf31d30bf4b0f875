#!/bin/bash
# Round 6, late: the backward's evidence set on the final binaries — PMC passes (separate runs, counters only) and rocprofv3 kernel stats of bench.py --mode bwd at config 3,
# the bench line itself, and the shapes table (same process, the library of the round's start is not needed: absolute numbers).  Lands in gpurun_out/r6fb/.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/r6fb
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for k in bwd_kv bwd_kernel; do
  KERNEL_FILTER=$k timeout 900 python tools/prof_pmc.py gpurun_out/r6fb/pmc_${k}_cfg3 -- python $R/bench.py --mode bwd --steps 10 --warmup 3 --precondition-s 0.3 --no-cpu-baseline > gpurun_out/r6fb/pmc_$k.log 2>&1
done
( cd /tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o k -- python $R/bench.py --mode bwd --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_b.log 2>&1;
  f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r6fb/kernel_stats_bwd_cfg3.csv; tail -1 /tmp/prof_b.log | grep -v amdgpu > $R/gpurun_out/r6fb/bench_bwd_under_rocprof.json )
python bench.py --mode bwd --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6fb/bench_bwd.json
timeout 900 python tools/ab_libs.py tiny-flash-attention_amd/lib_r6final/libtfa_hip.so tiny-flash-attention_amd/lib/libtfa_hip.so --bwd --variant 30 --cfgs cfg3,cfg3nc,cfg4,cfg5,gqa,gqanc,f16c,d64c --rounds 5 --iters 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r6fb/bwd_round_ab.txt
( echo "fuzz_bwd --n 300 --seed 7700:"; timeout 1500 python tools/fuzz_bwd.py --n 300 --seed 7700 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r6fb/fuzz_bwd.txt
grep -E "GRBM|mfma_pipe|valu_insts" gpurun_out/r6fb/pmc_bwd_kv_cfg3.txt gpurun_out/r6fb/pmc_bwd_kernel_cfg3.txt; head -3 gpurun_out/r6fb/kernel_stats_bwd_cfg3.csv | cut -c1-200; cut -c1-330 gpurun_out/r6fb/bench_bwd.json; cat gpurun_out/r6fb/bwd_round_ab.txt gpurun_out/r6fb/fuzz_bwd.txt
