#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests7.log
tail -15 gpurun_out/r4/gpu_tests7.log
true
