"""Interleaved A/B timing of kernel variants of the loaded library on the BASELINE shapes (same process, same inputs).
usage: python tools/ab_variants.py --variants 30,33 [--cfgs cfg3,cfg3nc,cfg4] [--rounds 5] [--iters 30]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg2": (4, 8, 1024, 64, torch.float16, False), "cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg4c": (1, 16, 16384, 128, torch.bfloat16, True), "cfg5": (8, 32, 4096, 128, torch.bfloat16, True),
       "d64": (4, 32, 4096, 64, torch.float16, False), "d64c": (4, 32, 4096, 64, torch.float16, True), "n2k": (8, 32, 2048, 128, torch.bfloat16, True),
       "n1k": (16, 32, 1024, 128, torch.bfloat16, True), "cfg3h16": (4, 16, 4096, 128, torch.bfloat16, True),
       "n8k": (2, 32, 8192, 128, torch.bfloat16, True), "cfg2b": (4, 8, 1024, 128, torch.bfloat16, False),
       "s2k": (1, 16, 2048, 128, torch.bfloat16, False), "s512": (8, 8, 512, 64, torch.float16, False), "s4k": (1, 8, 4096, 128, torch.bfloat16, False),
       "c8_4k": (1, 8, 4096, 128, torch.bfloat16, True), "c16_4k": (1, 16, 4096, 128, torch.bfloat16, True), "c8_8k": (1, 8, 8192, 128, torch.bfloat16, True),
       "c8_2k": (1, 8, 2048, 128, torch.bfloat16, True), "c16_2k": (1, 16, 2048, 128, torch.bfloat16, True), "c32_2k": (1, 32, 2048, 128, torch.bfloat16, True),
       "c32_1k": (1, 32, 1024, 128, torch.bfloat16, True), "c64_512": (1, 64, 512, 128, torch.bfloat16, True), "c16_1k": (1, 16, 1024, 128, torch.bfloat16, True),
       "s16_4k": (1, 16, 4096, 128, torch.bfloat16, False), "s32_2k": (1, 32, 2048, 128, torch.bfloat16, False), "s8_8k": (1, 8, 8192, 128, torch.bfloat16, False),
       "c2x16_2k": (2, 16, 2048, 64, torch.float16, True), "c4_16k": (1, 4, 16384, 128, torch.bfloat16, True),
       "c32_4k": (1, 32, 4096, 128, torch.bfloat16, True), "c2x16_4k": (2, 16, 4096, 64, torch.float16, True)}
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="30,33")
ap.add_argument("--cfgs", default="cfg3,cfg3nc,cfg4")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--warm", type=float, default=1.0, help="seconds of pre-conditioning launches before the rounds")
ap.add_argument("--dbgs", default="", help="comma list of tfa_debug_set_flags values: the A/B axis becomes (first variant, flags) instead of variants")
ap.add_argument("--check", action="store_true", help="compare every arm's output bits with the first arm's")
ap.add_argument("--data", default="normal", choices=["normal", "zeros", "ones", "small"],
                help="input values: the reference's normal(0,0.5), all zeros, all ones, or normal(0,0.01) — same instruction stream, "
                     "different switching activity (the DVFS give-back experiment of MI355X_MICROARCH.md)")
a = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()
vs = [int(x) for x in a.variants.split(",")]
arms = [(vs[0], int(f)) for f in a.dbgs.split(",")] if a.dbgs else [(v, 0) for v in vs]
import time
for cfg in a.cfgs.split(","):
    B, H, N, D, dt, causal = CFG[cfg]
    if a.data == "normal":
        mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
    elif a.data == "small":
        mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.01).to(dt)
    elif a.data == "zeros":
        mk = lambda: torch.zeros((B, H, N, D), dtype=dt, device=dev)
    else:
        mk = lambda: torch.ones((B, H, N, D), dtype=dt, device=dev)
    q, k, v = mk(), mk(), mk()
    out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl, by = C.c_double(), C.c_double()
    L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
    ms = C.c_float()
    t0 = time.time()
    _lib.set_variant(vs[0])
    while time.time() - t0 < a.warm:
        _lib.check(L.tfa_fwd_time(C.byref(p), 0, 50, s, C.byref(ms)))
    res = {arm: [] for arm in arms}
    for r in range(a.rounds):
        for arm in arms:
            _lib.set_variant(arm[0]); _lib.debug_set_flags(arm[1])
            _lib.check(L.tfa_fwd_time(C.byref(p), 3, a.iters, s, C.byref(ms)))
            res[arm].append(fl.value / (ms.value * 1e-3) / 1e12)
    same = ""
    if a.check:
        outs = []
        for arm in arms:
            _lib.set_variant(arm[0]); _lib.debug_set_flags(arm[1])
            out.zero_(); lse.zero_()
            _lib.check(L.tfa_fwd_time(C.byref(p), 0, 1, s, C.byref(ms)))
            torch.cuda.synchronize()
            outs.append((out.clone(), lse.clone()))
        same = " bits: " + " ".join("same" if torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) else "DIFF" for o in outs)
    _lib.set_variant(-1); _lib.debug_set_flags(0)
    name = lambda arm: f"v{arm[0]}" + (f"/dbg{arm[1]:#x}" if a.dbgs else "")
    print(cfg, f"[{a.data}]", " ".join(f"{name(arm)}: med {sorted(res[arm])[len(res[arm]) // 2]:7.1f} max {max(res[arm]):7.1f} TF" for arm in arms) + same, flush=True)
