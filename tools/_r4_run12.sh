#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_f32_gpu.py -m gpu -q > gpurun_out/r4/gpu_tests12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests12.log
tail -25 gpurun_out/r4/gpu_tests12.log
