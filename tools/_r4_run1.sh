#!/bin/bash
# round 4, GPU call 1: TAIL / PREF2 arms — bit identity, interleaved A/B, causal pass anatomy
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 300 python tools/r4_bits.py --base 30 --arms 38,39,40 > gpurun_out/r4/bits1.txt 2>&1; echo "bits rc=$?" >> gpurun_out/r4/bits1.txt
timeout 300 python tools/ab_variants.py --variants 30,38,39,40 --cfgs cfg3,cfg3nc,cfg4,n2k --rounds 7 --iters 40 --check > gpurun_out/r4/ab1.txt 2>&1
timeout 200 python tools/ab_variants.py --variants 30,38,39,40 --cfgs cfg3,cfg3nc --rounds 5 --iters 40 --data zeros >> gpurun_out/r4/ab1.txt 2>&1
for v in 30 40; do
  timeout 120 python tools/trace_passes.py $v >> gpurun_out/r4/anatomy1.txt 2>&1
  timeout 120 python tools/trace_passes.py $v pass1 >> gpurun_out/r4/anatomy1.txt 2>&1
done
tail -3 gpurun_out/r4/bits1.txt; cat gpurun_out/r4/ab1.txt; cat gpurun_out/r4/anatomy1.txt
