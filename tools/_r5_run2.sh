#!/bin/bash
# GPU call 2 (round 5): the exact-il8 instantiation (variant 38): parity, then speed beside variants 30 (lazy) and 17 (burst, exact)
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_abi.py -x -q -m gpu -k "exact or 38 or product_kernels" 2>&1 | tail -15
  echo "== speed"; timeout 300 python tools/ab_variants.py --variants 30,38,17 --cfgs cfg3,cfg3nc,cfg4 --rounds 5 --iters 30
  echo "== zeros"; timeout 300 python tools/ab_variants.py --variants 30,38,17 --cfgs cfg3,cfg3nc --rounds 5 --iters 30 --data zeros ) > gpurun_out/r5_exact_il8.txt 2>&1
tail -40 gpurun_out/r5_exact_il8.txt
