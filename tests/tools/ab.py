"""A/B of kernel variants on the GPU box: quick correctness of each variant against the oracle on a
small problem, then interleaved timing rounds on the BASELINE shapes.
usage: python tests/tools/ab.py [--variants 1,4,5] [--cfgs cfg3,cfg3nc] [--rounds 3]"""
import argparse
import ctypes as C
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg2": (4, 8, 1024, 64, torch.float16, False),
       "cfg2c": (4, 8, 1024, 64, torch.float16, True), "cfg5": (8, 32, 4096, 128, torch.bfloat16, True),
       "d64": (4, 32, 4096, 64, torch.float16, False), "d64c": (4, 32, 4096, 64, torch.float16, True),
       "n2k": (8, 32, 2048, 128, torch.bfloat16, True), "n1k": (16, 32, 1024, 128, torch.bfloat16, True),
       "n2knc": (8, 32, 2048, 128, torch.bfloat16, False), "n1knc": (16, 32, 1024, 128, torch.bfloat16, False),
       "n512": (32, 32, 512, 128, torch.bfloat16, True), "n512nc": (32, 32, 512, 128, torch.bfloat16, False),
       "n8k": (2, 32, 8192, 128, torch.bfloat16, True),
       "s1": (1, 8, 2048, 128, torch.bfloat16, True), "s2": (2, 16, 1024, 128, torch.bfloat16, False), "s3": (1, 32, 4096, 128, torch.bfloat16, True)}
dev = torch.device("cuda:0")


def quick_check(variant):
    ok = True
    for (dt, B, H, N, D, causal, Hk, Nk) in ((torch.bfloat16, 1, 2, 712, 128, True, None, None), (torch.float16, 1, 4, 300, 64, False, 2, 450),
                                               (torch.bfloat16, 1, 2, 512, 128, False, None, None), (torch.float16, 2, 2, 520, 64, True, None, None)):
        q, k, v = O.make_inputs(B, H, N, D, dt, seed=3, Hk=Hk, Nk=Nk)
        sc = 1 / math.sqrt(D)
        emu = O.tiled_emulation_lazy if _lib.variant_name(variant).startswith('il') else O.tiled_emulation
        ref, lref = emu(q, k, v, causal, sc, 64, return_lse=True)
        _lib.set_variant(variant)
        o32, lse = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc, out_f32=True)
        o16, _ = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc)
        torch.cuda.synchronize()
        e32 = (o32.cpu() - ref).abs().max().item()
        e16 = (o16.float().cpu() - ref).abs().max().item()
        el = (lse.cpu() - lref).abs().max().item()
        good = e32 < 2e-4 and e16 < 8e-3 and el < 1e-4
        ok &= good
        if not good:
            print(f"  CHECK FAIL variant {variant} {dt} N{N} D{D} causal={causal}: e32={e32:.2e} e16={e16:.2e} lse={el:.2e}")
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="")
    ap.add_argument("--cfgs", default="cfg3,cfg3nc")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    variants = [int(x) for x in a.variants.split(",")] if a.variants else list(range(_lib.num_variants()))
    good = {v: quick_check(v) for v in variants}
    print("check:", {v: ("ok" if g else "FAIL") for v, g in good.items()})
    res = {}
    for cfg in a.cfgs.split(","):
        B, H, N, D, dt, causal = CFG[cfg]
        mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
        q, k, v = mk(), mk(), mk()
        out = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
        fl = C.c_double()
        _lib.lib().tfa_fwd_work(C.byref(p), C.byref(fl), None)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for r in range(a.rounds):
            for var in variants:
                _lib.set_variant(var)
                ms = C.c_float()
                _lib.check(_lib.lib().tfa_fwd_time(C.byref(p), 3, a.iters, s, C.byref(ms)))
                res.setdefault((cfg, var), []).append(fl.value / (ms.value * 1e-3) / 1e12)
        for var in variants:
            tf = sorted(res[(cfg, var)])
            print(f"{cfg:7s} v{var:<2d} {tf[len(tf)//2]:7.1f} TF (min {tf[0]:.1f} max {tf[-1]:.1f})  {tf[len(tf)//2]/25:.1f}%  {_lib.variant_name(var)}")
    _lib.set_variant(-1)


if __name__ == "__main__":
    main()
