#!/bin/bash
# Round 6: the evidence set of the final binaries — GPU tests, smoke, the bench line under the driver's command, rocprofv3 kernel stats of the same
# command, PMC passes (separate runs, counters only) for configs 3 and 4 and the exact-max twin, the causal pass anatomy, the diagonal's per-wave trace.
# Everything lands in gpurun_out/r6f/ and is copied into profiles/r06_* afterwards (tools/update_hbm_traffic.py stamps the traffic entries).
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/r6f
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r6f/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/r6f/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r6f/bench.err | grep -v amdgpu | tail -1 > gpurun_out/r6f/bench_driver_protocol.json
( cd /tmp && rm -rf /tmp/prof_k && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_k.log 2>&1;
  f=$(find /tmp/prof_k -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r6f/kernel_stats_bench_cfg3.csv; tail -1 /tmp/prof_k.log | grep -v amdgpu > $R/gpurun_out/r6f/bench_under_rocprof.json )
for cfg in cfg3 cfg4; do
  KERNEL_FILTER=fwd_kernel timeout 900 python tools/prof_pmc.py gpurun_out/r6f/pmc_default_$cfg -- python $R/bench.py --config $cfg --steps 10 --warmup 3 --precondition-s 0.3 --no-cpu-baseline --no-secondary > gpurun_out/r6f/pmc_$cfg.log 2>&1
done
KERNEL_FILTER=fwd_kernel timeout 900 python tools/prof_pmc.py gpurun_out/r6f/pmc_exact_cfg3 -- python $R/tools/run_cfg.py cfg3 --n 40 --variant 38 > gpurun_out/r6f/pmc_exact.log 2>&1
for a in "" pass1; do timeout 120 python tools/trace_passes.py 30 $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6f/causal_pass_anatomy.txt; done
timeout 300 python tools/ab_variants.py --variants 30 --cfgs cfg3,cfg3nc,cfg4,cfg5,cfg2 --rounds 5 --iters 40 2>&1 | grep -v amdgpu > gpurun_out/r6f/shapes.txt
timeout 200 python tools/ab_variants.py --variants 30 --cfgs cfg3,cfg3nc,cfg4 --rounds 5 --iters 40 --data zeros 2>&1 | grep -v amdgpu >> gpurun_out/r6f/shapes.txt
timeout 300 python tools/ab_variants.py --variants 30,38 --cfgs cfg3,cfg3nc,cfg4 --rounds 5 --iters 40 2>&1 | grep -v amdgpu > gpurun_out/r6f/exact_vs_default.txt
sha256sum tiny-flash-attention_amd/lib/libtfa_hip.so > gpurun_out/r6f/lib.sha256
cat gpurun_out/r6f/gpu_tests.log gpurun_out/r6f/smoke.log gpurun_out/r6f/shapes.txt; head -5 gpurun_out/r6f/kernel_stats_bench_cfg3.csv | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6f/bench_driver_protocol.json'))
print('HEADLINE', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_event_clock'), 'cpu', d['cpu_baseline']['value'], d.get('value_at_reference_rounding_points', {}).get('value'),
      {k:(round(v.get('tflops',0),1), round(v.get('ms',0),4)) for k,v in d['secondary'].items()})
PY
