#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_splitkv_gpu.py tests/test_dist_gpu.py -m gpu -q -k "exact_running_max or product_kernels or splitkv_matches or bench" > gpurun_out/r4/gpu_tests8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests8.log
tail -6 gpurun_out/r4/gpu_tests8.log
