export TMPDIR=/tmp
(rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | head -60) > gpurun_out/r03_smi_idle.txt
python tools/power_trace.py --cfg cfg3 --seconds 3 --out gpurun_out/r03_power_trace_cfg3.csv 2>&1 | grep -v amdgpu
python tools/power_trace.py --cfg cfg4 --seconds 3 --data normal,zeros --out gpurun_out/r03_power_trace_cfg4.csv 2>&1 | grep -v amdgpu
python tools/bench_bwd.py --cfgs cfg3 2>&1 | grep -v amdgpu | tee gpurun_out/r03_bwd_baseline.txt
