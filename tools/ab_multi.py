"""Interleaved A/B of several BUILDS of the library in one process (same inputs, same box, round-robin):
usage: python tools/ab_multi.py name=path[:variant] ... [--cfgs cfg3,cfg4] [--rounds 5] [--iters 30]
Each entry loads its own copy of libtfa_hip.so with ctypes; `variant` defaults to 33 (the x4 kernel)."""
import argparse, ctypes as C, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg2": (4, 8, 1024, 64, torch.float16, False), "cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg4c": (1, 16, 16384, 128, torch.bfloat16, True), "cfg5": (8, 32, 4096, 128, torch.bfloat16, True),
       "n2k": (8, 32, 2048, 128, torch.bfloat16, True), "n1k": (16, 32, 1024, 128, torch.bfloat16, True), "d64": (4, 32, 4096, 64, torch.float16, False),
       "d64c": (4, 32, 4096, 64, torch.float16, True), "decode": (64, 32, 1, 128, torch.bfloat16, True, 8, 8192), "decmha": (64, 32, 1, 128, torch.bfloat16, True, 32, 8192),
       "d192c": (4, 16, 4096, 192, torch.bfloat16, True), "d160c": (4, 16, 4096, 160, torch.bfloat16, True), "d224nc": (4, 8, 4096, 224, torch.float16, False),
       "d256c": (4, 8, 4096, 256, torch.bfloat16, True), "d256nc": (4, 8, 4096, 256, torch.bfloat16, False), "d256f16c": (4, 8, 4096, 256, torch.float16, True),
       "d256n16k": (1, 16, 16384, 256, torch.bfloat16, False), "d64bfc": (4, 32, 4096, 64, torch.bfloat16, True), "n512": (32, 32, 512, 128, torch.bfloat16, True),
       "n512d64": (32, 16, 512, 64, torch.float16, False),
       # grids of at most one / two 128-row blocks per CU (the key-split rules of pick_variant)
       "k1": (1, 16, 2048, 128, torch.bfloat16, False), "k2": (1, 8, 4096, 128, torch.bfloat16, False), "k3": (4, 8, 1024, 128, torch.bfloat16, False),
       "k4": (1, 8, 4096, 128, torch.bfloat16, True), "k5": (1, 16, 2048, 128, torch.bfloat16, True), "k6": (1, 64, 512, 128, torch.bfloat16, True),
       "k7": (2, 8, 2048, 64, torch.float16, False), "k8": (1, 8, 8192, 64, torch.float16, False), "k9": (4, 8, 1024, 64, torch.float16, True),
       "k10": (1, 32, 1024, 128, torch.bfloat16, False), "k11": (2, 8, 1024, 64, torch.float16, False), "k12": (1, 16, 1024, 128, torch.bfloat16, True),
       "p1": (1, 16, 4096, 128, torch.bfloat16, True), "p2": (1, 8, 8192, 128, torch.bfloat16, True), "p3": (1, 4, 16384, 128, torch.bfloat16, True),
       "p4": (2, 16, 4096, 64, torch.float16, True), "p5": (1, 32, 4096, 128, torch.bfloat16, True), "f16c": (4, 16, 4096, 128, torch.float16, True), "f16nc": (4, 16, 4096, 128, torch.float16, False)}
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--cfgs", default="cfg3,cfg3nc,cfg4")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--warm", type=float, default=1.0)
ap.add_argument("--check", action="store_true", help="also compare every build's output bits with the first build's")
ap.add_argument("--data", default="normal", choices=["normal", "zeros"], help="normal(0, 0.5) = the reference's recipe (power-capped), zeros = cycle-bound")
a = ap.parse_args()
dev = torch.device("cuda:0")
P = C.POINTER(_lib.TfaFwdParams)
entries = []
for spec in a.libs:
    name, rest = spec.split("=", 1)
    path, _, var = rest.partition(":")
    L = C.CDLL(os.path.abspath(path))
    L.tfa_fwd_time.argtypes = [P, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    L.tfa_set_variant.argtypes = [C.c_int]
    entries.append((name, L, int(var) if var else 33))
for cfg in a.cfgs.split(","):
    B, H, N, D, dt, causal = CFG[cfg][:6]
    Hk, Nk = (CFG[cfg][6], CFG[cfg][7]) if len(CFG[cfg]) > 6 else (H, N)      # (decode-like entries: K/V heads and keys differ from the query side)
    mk = lambda h=H, n=N: (torch.zeros((B, h, n, D), dtype=dt, device=dev) if a.data == "zeros" else torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt))
    q, k, v = mk(), mk(Hk, Nk), mk(Hk, Nk)
    out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl, by = C.c_double(), C.c_double()
    _lib.lib().tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
    ms = C.c_float()
    t0 = time.time()
    name, L, var = entries[0]
    assert L.tfa_set_variant(var) == 0, (name, var)
    while time.time() - t0 < a.warm:
        assert L.tfa_fwd_time(C.byref(p), 0, 50, s, C.byref(ms)) == 0
    res = {e[0]: [] for e in entries}
    for r in range(a.rounds):
        for name, L, var in entries:
            assert L.tfa_set_variant(var) == 0, (name, var)
            st = L.tfa_fwd_time(C.byref(p), 3, a.iters, s, C.byref(ms))
            assert st == 0, (name, st)
            res[name].append(fl.value / (ms.value * 1e-3) / 1e12)
    same = ""
    if a.check:
        outs = []
        for name, L, var in entries:
            L.tfa_set_variant(var)
            out.zero_(); lse.zero_()
            assert L.tfa_fwd_time(C.byref(p), 0, 1, s, C.byref(ms)) == 0
            torch.cuda.synchronize()
            outs.append((out.clone(), lse.clone()))
        same = "  bits: " + " ".join(f"{e[0]}={'same' if torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) else 'DIFF'}" for e, o in zip(entries, outs))
    unit = 1.0
    if cfg.startswith("dec"):                              # decode: report the K/V streaming rate in TB/s instead of TFLOP/s
        unit = by.value / fl.value
    print(f"{cfg:7s}", "  ".join(f"{n}: {sorted(x)[len(x) // 2] * unit:7.{3 if unit != 1.0 else 1}f}" for n, x in res.items()) + same + ("  (TB/s)" if unit != 1.0 else ""), flush=True)
