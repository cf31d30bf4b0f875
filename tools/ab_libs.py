"""A/B of two BUILDS of the library in one process (interleaved rounds, same inputs, same box):
usage: python tools/ab_libs.py LIB_A LIB_B [--variant 28] [--cfgs cfg3,cfg4] [--rounds 5] [--iters 40]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg5": (8, 32, 4096, 128, torch.bfloat16, True),
       "d64": (4, 32, 4096, 64, torch.float16, False), "d64c": (4, 32, 4096, 64, torch.float16, True),
       "gqa": (4, 32, 4096, 128, torch.bfloat16, True, 8), "gqanc": (4, 32, 4096, 128, torch.bfloat16, False, 8), "f16c": (4, 32, 4096, 128, torch.float16, True)}
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs=2)
ap.add_argument("--variant", type=int, default=28)
ap.add_argument("--variant-b", type=int, default=None)
ap.add_argument("--cfgs", default="cfg3,cfg4")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--bwd", action="store_true", help="time tfa_bwd instead of tfa_fwd")
a = ap.parse_args()
dev = torch.device("cuda:0")
P = C.POINTER(_lib.TfaFwdParams)
Ls = []
for path in a.libs:
    L = C.CDLL(os.path.abspath(path))
    L.tfa_fwd_time.argtypes = [P, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    L.tfa_set_variant.argtypes = [C.c_int]
    if a.bwd:
        L.tfa_bwd_time.argtypes = [C.POINTER(_lib.TfaBwdParams), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    Ls.append(L)
va = a.variant
vb = a.variant if a.variant_b is None else a.variant_b
for cfg in a.cfgs.split(","):
    B, H, N, D, dt, causal = CFG[cfg][:6]
    Hk = CFG[cfg][6] if len(CFG[cfg]) > 6 else H          # (a seventh entry: K/V heads — GQA)
    mk = lambda h=H: torch.empty((B, h, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
    q, k, v = mk(), mk(Hk), mk(Hk)
    out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl, by = C.c_double(), C.c_double()
    _lib.lib().tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
    if a.bwd:
        _lib.check(_lib.lib().tfa_fwd(C.byref(p), s))
        dout = mk()
        dq, dk, dv, delta = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(lse)
        pb = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, 1 / math.sqrt(D))
        _lib.lib().tfa_bwd_work(C.byref(pb), C.byref(fl), C.byref(by))
    res = [[], []]
    for r in range(a.rounds + 1):
        for i, (L, var) in enumerate(zip(Ls, (va, vb))):
            assert L.tfa_set_variant(var) == 0
            ms = C.c_float()
            st = L.tfa_bwd_time(C.byref(pb), 2, a.iters, s, C.byref(ms)) if a.bwd else L.tfa_fwd_time(C.byref(p), 2, a.iters, s, C.byref(ms))
            assert st == 0, st
            if r:
                res[i].append(fl.value / (ms.value * 1e-3) / 1e12)
    ma, mb = sorted(res[0])[len(res[0]) // 2], sorted(res[1])[len(res[1]) // 2]
    print(f"{cfg:7s} A(v{va}) median {ma:7.1f} TF [{min(res[0]):7.1f}..{max(res[0]):7.1f}]   B(v{vb}) median {mb:7.1f} TF [{min(res[1]):7.1f}..{max(res[1]):7.1f}]   B/A = {mb / ma:.4f}")
