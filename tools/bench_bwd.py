"""Backward timing on the BASELINE shapes: tfa_bwd_time (delta + dQ + dK + dV kernels per call).
TFLOP/s uses the conventional 2.5 x forward flops (5 GEMMs); the kernels execute 7 GEMM units (dQ launch: S, dP, dQ; fused dK/dV launch: S, dP, dV, dK).
usage: python tools/bench_bwd.py [--cfgs cfg3,cfg3nc,cfg4] [--iters 20]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg2": (4, 8, 1024, 64, torch.float16, False),
       "cfg5": (8, 32, 4096, 128, torch.bfloat16, True),
       # does the dS workspace pay when it fits the 256 MB memory-side cache?  dS = B*H*N*N*2 bytes: 134 MB, 268 MB, 1.07 GB
       "m1": (1, 16, 2048, 128, torch.bfloat16, False), "m2": (2, 16, 2048, 128, torch.bfloat16, False), "m8": (8, 16, 2048, 128, torch.bfloat16, False),
       "n1": (1, 64, 1024, 128, torch.bfloat16, False), "n8": (8, 64, 1024, 128, torch.bfloat16, False),
       # head dims above 128 (one wave per SIMD, three single-gradient launches)
       "d256c": (4, 8, 4096, 256, torch.bfloat16, True), "d256": (4, 8, 4096, 256, torch.bfloat16, False), "d192c": (4, 16, 4096, 192, torch.bfloat16, True),
       "d256h": (4, 8, 4096, 256, torch.float16, True), "d160c": (4, 16, 4096, 160, torch.bfloat16, True), "d224c": (4, 8, 4096, 224, torch.bfloat16, True),
       "d96c": (4, 32, 4096, 96, torch.bfloat16, True), "d80c": (4, 32, 4096, 80, torch.bfloat16, True), "d32c": (4, 32, 4096, 32, torch.float16, True)}
ap = argparse.ArgumentParser()
ap.add_argument("--cfgs", default="cfg3,cfg3nc,cfg4,cfg2")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
for cfg in a.cfgs.split(","):
    B, H, N, D, dt, causal = CFG[cfg]
    mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
    q, k, v, dout = mk(), mk(), mk(), mk()
    sc = 1 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(q, k, v, causal, sc)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse)
    p = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl, by = C.c_double(), C.c_double()
    _lib.check(_lib.lib().tfa_bwd_work(C.byref(p), C.byref(fl), C.byref(by)))
    best, best_split, best_ws = 1e9, 1e9, 1e9
    need = ops.bwd_workspace_bytes(p)
    ws = torch.empty((need,), dtype=torch.uint8, device=dev) if need > 0 else None
    for r in range(3):
        ms = C.c_float()
        _lib.check(_lib.lib().tfa_bwd_time(C.byref(p), 3, a.iters, s, C.byref(ms)))
        best = min(best, ms.value)
        _lib.debug_bwd_split(True)                     # the 8-GEMM form: dK and dV as two launches
        try:
            _lib.check(_lib.lib().tfa_bwd_time(C.byref(p), 3, a.iters, s, C.byref(ms)))
        finally:
            _lib.debug_bwd_split(False)
        best_split = min(best_split, ms.value)
        if ws is not None:                             # the 5-GEMM form: dS kept in a workspace, dQ = dS.K
            p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
            try:
                _lib.check(_lib.lib().tfa_bwd_time(C.byref(p), 3, a.iters, s, C.byref(ms)))
            finally:
                p.workspace, p.workspace_bytes = None, 0
            best_ws = min(best_ws, ms.value)
    pf = ops.make_params(q, k, v, out, lse, causal, sc)
    msf = C.c_float()
    _lib.check(_lib.lib().tfa_fwd_time(C.byref(pf), 3, a.iters, s, C.byref(msf)))
    print(f"{cfg:7s} bwd {best:7.3f} ms = {fl.value / best / 1e9:7.1f} TFLOP/s (2.5x-fwd convention; {fl.value / best / 1e9 * 1.4:7.1f} executed) "
          f"| fwd {msf.value:6.3f} ms | bwd/fwd = {best / msf.value:.2f} | algorithmic {by.value / 1e6:.0f} MB | two-launch dK,dV form: {best_split:7.3f} ms | with a {need / 1e9:.2f} GB dS workspace (5 GEMMs): {best_ws:7.3f} ms = {fl.value / best_ws / 1e9:7.1f} TFLOP/s")
