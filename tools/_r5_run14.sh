#!/bin/bash
# GPU call 14 (round 5): final evidence set — PMC passes (cfg3 default + exact, cfg4), the driver's bench command twice, rocprofv3 kernel stats (csv),
# the in-library MFMA-only probe BEFORE and AFTER the stand-alone one (order effect)
cd /root/repo; mkdir -p gpurun_out
B="python /root/repo/bench.py --config cfg3 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --precondition-s 0.5"
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_default_cfg3 -- $B ) > gpurun_out/r05_pmc_default.log 2>&1
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_exact_cfg3 -- $B --variant 38 ) > gpurun_out/r05_pmc_exact.log 2>&1
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_default_cfg4 -- python /root/repo/bench.py --config cfg4 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --precondition-s 0.5 ) > gpurun_out/r05_pmc_default4.log 2>&1
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_a.json 2> gpurun_out/r05_bench_driver_a.err
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r05_bench_driver_b.json 2> gpurun_out/r05_bench_driver_b.err
cd /tmp && export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5_prof3 -o r5 -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline ) > /root/repo/gpurun_out/r5_prof3.log 2>&1
cd /root/repo
cat > /tmp/probe.py <<'PY'
import ctypes as C, torch, sys
sys.path.insert(0, "/root/repo")
from tiny_flash_attention_amd import _lib
L = _lib.lib()
q = torch.empty((4, 32, 4096, 128), dtype=torch.float32, device="cuda").normal_(0, 0.5).to(torch.bfloat16)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(2):
    t = C.c_double()
    print("in-library", L.tfa_debug_mfma_ceiling(C.c_void_p(q.data_ptr()), C.c_ulonglong(q.numel() * 2), C.c_double(2.0), s, C.byref(t)), round(t.value, 1), "TF", flush=True)
PY
( echo "== in-library probe first (cold process)"; timeout 120 python /tmp/probe.py; echo "== tools/probe_mfma_power 2.0"; timeout 120 tools/probe_mfma_power 2.0 | grep -E "normal.*breuse=2$|sustained" | head -6; echo "== in-library probe again"; timeout 120 python /tmp/probe.py ) > gpurun_out/r05_mfma_ceiling_crosscheck2.txt 2>&1
grep -E "busy|per_mfma" gpurun_out/r05_pmc_default_cfg3.txt gpurun_out/r05_pmc_exact_cfg3.txt gpurun_out/r05_pmc_default_cfg4.txt; python3 -c "
import json
for f in ('a','b'):
    j=json.load(open('gpurun_out/r05_bench_driver_%s.json'%f)); r=j['roofline']; print(f, round(j['value'],1), round(r['frac'],4), r['mfma_only_ceiling_random_data'], r['frac_of_mfma_only_ceiling'], j['box']['gpu_id'])
    for k,v in j.get('secondary',{}).items(): print('   ',k, round(v.get('ms',0),4), round(v.get('tflops',0),1), round(v.get('frac',0),4), round(v.get('hbm_frac',0),3))
"; grep -v amdgpu.ids gpurun_out/r05_mfma_ceiling_crosscheck2.txt; head -3 gpurun_out/r5_prof3/r5_kernel_stats.csv | cut -c1-160
