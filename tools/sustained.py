"""Burst vs sustained throughput per variant: consecutive chunks of back-to-back launches (no idle gaps), so the
power-managed clock settles.  usage: python tools/sustained.py [--variants 17,19,28] [--cfg cfg3] [--chunks 8] [--iters 100]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg5": (8, 32, 4096, 128, torch.bfloat16, True)}
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="17,19,28")
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--chunks", type=int, default=8)
ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, H, N, D, dt, causal = CFG[a.cfg]
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fl, by = C.c_double(), C.c_double()
_lib.lib().tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
for var in [int(x) for x in a.variants.split(",")]:
    _lib.set_variant(var)
    torch.cuda.synchronize()
    res = []
    for c in range(a.chunks):
        ms = C.c_float()
        _lib.check(_lib.lib().tfa_fwd_time(C.byref(p), 0 if c else 2, a.iters, s, C.byref(ms)))
        res.append(fl.value / (ms.value * 1e-3) / 1e12)
    print(f"{a.cfg} v{var:2d} TF per chunk of {a.iters}: " + " ".join(f"{r:7.1f}" for r in res) + f"   | {_lib.variant_name(var)[:40]}")
_lib.set_variant(-1)
