"""Driven by tools/power_trace.py --cmd: runs the backward of BASELINE shapes back to back for a few seconds each, printing ARM_BEGIN / ARM_END lines.
usage: python tools/power_trace.py --cmd "python tools/power_bwd_cmd.py" --out gpurun_out/power_bwd.csv"""
import ctypes as C, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False), "cfg4": (1, 16, 16384, 128, torch.bfloat16, False)}
dev = torch.device("cuda:0")
L = _lib.lib()
L.tfa_bwd_time.argtypes = [C.POINTER(_lib.TfaBwdParams), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
for cfg in ("cfg3", "cfg3nc", "cfg4"):
    for data in ("normal", "zeros"):
        B, H, N, D, dt, causal = CFG[cfg]
        mk = (lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)) if data == "normal" else (lambda: torch.zeros((B, H, N, D), dtype=dt, device=dev))
        q, k, v, dout = mk(), mk(), mk(), mk()
        sc = 1 / math.sqrt(D)
        out, lse = ops.flash_attn_fwd(q, k, v, causal, sc)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        pb = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ms = C.c_float()
        _lib.check(L.tfa_bwd_time(C.byref(pb), 2, 20, s, C.byref(ms)))
        time.sleep(0.5)
        print(f"ARM_BEGIN bwd {cfg} {data}", flush=True)
        t0 = time.perf_counter()
        tot, n = 0.0, 0
        while time.perf_counter() - t0 < 3.0:
            _lib.check(L.tfa_bwd_time(C.byref(pb), 0, 50, s, C.byref(ms)))
            tot += ms.value; n += 1
        print(f"ARM_END {tot / n:.4f} ms per backward", flush=True)
