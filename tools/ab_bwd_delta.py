"""A/B of tfa_bwd with delta computed inside the dQ launch (default) against the launch of its own (tfa_debug_bwd_split bit 3), same process:
usage: python tools/ab_bwd_delta.py [--cfgs cfg3,cfg3nc,cfg4,d64c,d256c] [--rounds 5] [--iters 20]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False), "cfg4": (1, 16, 16384, 128, torch.bfloat16, False),
       "d64c": (4, 32, 4096, 64, torch.float16, True), "d256c": (4, 8, 4096, 256, torch.bfloat16, True), "n1k": (16, 32, 1024, 128, torch.bfloat16, True)}
ap = argparse.ArgumentParser()
ap.add_argument("--cfgs", default="cfg3,cfg3nc,cfg4,d64c,d256c,n1k")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()
L.tfa_bwd_time.argtypes = [C.POINTER(_lib.TfaBwdParams), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
for cfg in a.cfgs.split(","):
    B, H, N, D, dt, causal = CFG[cfg]
    mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
    q, k, v, dout = mk(), mk(), mk(), mk()
    sc = 1 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(q, k, v, causal, sc)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse)
    pb = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {0: [], 8: []}
    for r in range(a.rounds):
        for flag in (0, 8):
            _lib.debug_bwd_split(flag)
            ms = C.c_float()
            _lib.check(L.tfa_bwd_time(C.byref(pb), 2, a.iters, s, C.byref(ms)))
            res[flag].append(ms.value)
    _lib.debug_bwd_split(0)
    m0, m8 = sorted(res[0])[len(res[0]) // 2], sorted(res[8])[len(res[8]) // 2]
    print(f"{cfg:7s} delta inside the dQ launch: {m0:.4f} ms   launch of its own: {m8:.4f} ms   ratio {m8 / m0:.4f}", flush=True)
