#!/bin/bash
# GPU call 5 (round 5): PMC passes (default il8 and exact-il8 on the headline), the in-library MFMA-only probe beside tools/probe_mfma_power on one box, fuzz of the big grids
cd /root/repo; mkdir -p gpurun_out
B="python /root/repo/bench.py --config cfg3 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --precondition-s 0.5"
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_default_cfg3 -- $B ) > gpurun_out/r05_pmc_default.log 2>&1
( timeout 600 python tools/prof_pmc.py gpurun_out/r05_pmc_exact_cfg3 -- $B --variant 38 ) > gpurun_out/r05_pmc_exact.log 2>&1
( echo "== tools/probe_mfma_power 2.0"; timeout 120 tools/probe_mfma_power 2.0
  echo "== tfa_debug_mfma_ceiling, 3 calls of 2 s on a normal(0,0.5) bf16 tensor"
  timeout 120 python - <<'PY'
import ctypes as C, torch, sys
sys.path.insert(0, "/root/repo")
from tiny_flash_attention_amd import _lib
L = _lib.lib()
q = torch.empty((4, 32, 4096, 128), dtype=torch.float32, device="cuda").normal_(0, 0.5).to(torch.bfloat16)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(3):
    t = C.c_double()
    print(L.tfa_debug_mfma_ceiling(C.c_void_p(q.data_ptr()), C.c_ulonglong(q.numel() * 2), C.c_double(2.0), s, C.byref(t)), round(t.value, 1), "TF")
z = torch.zeros_like(q)
t = C.c_double(); L.tfa_debug_mfma_ceiling(C.c_void_p(z.data_ptr()), C.c_ulonglong(z.numel() * 2), C.c_double(2.0), s, C.byref(t)); print("zeros", round(t.value, 1), "TF")
PY
) > gpurun_out/r05_mfma_ceiling_crosscheck.txt 2>&1
( timeout 600 python tools/fuzz_fwd.py --big --n 250 --seed 5100 2>&1 | tail -5 ) > gpurun_out/r05_fuzz_big.txt 2>&1
cat gpurun_out/r05_pmc_default_cfg3.txt | head -40; cat gpurun_out/r05_mfma_ceiling_crosscheck.txt | grep -v amdgpu.ids | tail -25; cat gpurun_out/r05_fuzz_big.txt
