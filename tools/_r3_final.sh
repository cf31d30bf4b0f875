export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4) 2>&1 | tee gpurun_out/r03_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -9 | tee gpurun_out/r03_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r03_bench_final.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], {k:round(v.get('tflops',0),1) for k,v in d['secondary'].items()})"
