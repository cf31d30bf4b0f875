#!/bin/bash
# GPU call 16 (round 5): final validation — smoke, the whole GPU suite, the driver's bench command, the reference's classic configs
cd /root/repo; mkdir -p gpurun_out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 ) > gpurun_out/r05_smoke.log 2>&1; tail -3 gpurun_out/r05_smoke.log
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_gpu_tests_final.log 2>&1; tail -3 gpurun_out/r05_gpu_tests_final.log
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_final.json 2> gpurun_out/r05_bench_driver_final.err
python3 -c "
import json
j=json.load(open('gpurun_out/r05_bench_driver_final.json')); r=j['roofline']; print(round(j['value'],1), round(r['frac'],4), r['traffic'], round(r['mfma_only_ceiling_random_data'],1), round(r['frac_of_mfma_only_ceiling'],3), j['box']['gpu_id'])
for k,v in j.get('secondary',{}).items(): print('   ',k, round(v.get('ms',0),4), round(v.get('tflops',0),1), round(v.get('frac',0),4), round(v.get('hbm_frac',0),3))
"
( timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_classic_configs.txt; cat gpurun_out/r05_classic_configs.txt
