"""Bring-up / A-B script for the GPU box: parity of every kernel variant against the oracle on a
shape matrix, then timing of every variant on the BASELINE configs.  Writes gpurun_out/bringup.json.
(tools/ is developer tooling; the oracle is used here as the checker only.)"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tiny_flash_attention_amd as tfa  # noqa: E402
from tiny_flash_attention_amd import _lib, ops  # noqa: E402
from oracle import oracle as O  # noqa: E402
import ctypes as C  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
dev = torch.device("cuda:0")
res = {"parity": [], "timing": []}


def parity_case(variant, dtype, B, H, N, D, causal, Hk=None, Nk=None, seed=0):
    q, k, v = O.make_inputs(B, H, N, D, dtype, seed=seed, Hk=Hk, Nk=Nk)
    sc = 1.0 / math.sqrt(D)
    ref, lse_ref = O.exact64(q, k, v, causal, sc, p_round=dtype, return_lse=True)
    _lib.set_variant(variant)
    out, lse = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc)
    o32, _ = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc, out_f32=True)
    torch.cuda.synchronize()
    out = out.float().cpu(); o32 = o32.cpu(); lse = lse.cpu()
    fin = torch.isfinite(lse_ref)
    e16 = (out - ref).abs().max().item()
    e32 = (o32 - ref).abs().max().item()
    el = (lse[fin] - lse_ref[fin]).abs().max().item() if fin.any() else 0.0
    rel32 = ((o32 - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).max().item()
    ok = e16 < 1e-2 and e32 < 2e-3 and el < 1e-3 and bool(torch.isfinite(out).all())
    r = dict(variant=variant, dtype=str(dtype), B=B, H=H, N=N, D=D, causal=causal, Hk=Hk, Nk=Nk,
             err16=e16, err32=e32, rel32=rel32, err_lse=el, ok=ok)
    res["parity"].append(r)
    print(("PASS " if ok else "FAIL ") + json.dumps(r), flush=True)
    return ok


def time_case(variant, dtype, B, H, N, D, causal, iters=20):
    q = torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dtype)
    k = torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dtype)
    v = torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dtype)
    out = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, causal, 1.0 / math.sqrt(D))
    _lib.set_variant(variant)
    ms = C.c_float()
    st = _lib.lib().tfa_fwd_time(C.byref(p), 5, iters, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms))
    _lib.check(st)
    fl, by = C.c_double(), C.c_double()
    _lib.lib().tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
    tf = fl.value / (ms.value * 1e-3) / 1e12
    r = dict(variant=variant, name=_lib.variant_name(variant), dtype=str(dtype), B=B, H=H, N=N, D=D, causal=causal,
             ms=ms.value, tflops=tf, frac=tf / 2500.0)
    res["timing"].append(r)
    print("TIME " + json.dumps(r), flush=True)


def main():
    nv = _lib.num_variants()
    quick = [
        (torch.bfloat16, 1, 2, 256, 128, False), (torch.bfloat16, 1, 2, 256, 128, True),
        (torch.bfloat16, 2, 3, 512, 128, True), (torch.float16, 1, 2, 512, 64, False),
        (torch.float16, 2, 2, 320, 64, True), (torch.bfloat16, 1, 2, 200, 128, True),
        (torch.bfloat16, 1, 1, 77, 64, False), (torch.float16, 1, 2, 1024, 128, True),
    ]
    allok = True
    for var in range(nv):
        for (dt, B, H, N, D, c) in quick:
            try:
                allok &= parity_case(var, dt, B, H, N, D, c)
            except Exception as e:  # keep going: we want the whole matrix
                allok = False
                print("EXC", var, dt, B, H, N, D, c, repr(e), flush=True)
        # GQA + Nq != Nk
        for (Nq, Nk, c) in ((128, 384, True), (384, 128, True), (100, 333, False)):
            try:
                allok &= parity_case(var, torch.bfloat16, 1, 4, Nq, 128, c, Hk=2, Nk=Nk, seed=7)
            except Exception as e:
                allok = False
                print("EXC", var, Nq, Nk, c, repr(e), flush=True)
    res["all_parity_ok"] = allok
    for var in range(nv):
        try:
            time_case(var, torch.bfloat16, 4, 32, 4096, 128, True)     # cfg3 headline
            time_case(var, torch.bfloat16, 4, 32, 4096, 128, False)
            time_case(var, torch.float16, 4, 8, 1024, 64, False)       # cfg2
            time_case(var, torch.bfloat16, 1, 16, 16384, 128, False, iters=5)   # cfg4
        except Exception as e:
            print("EXC timing", var, repr(e), flush=True)
    _lib.set_variant(-1)
    json.dump(res, open(os.path.join(OUT, "bringup.json"), "w"), indent=1)
    print("ALL_PARITY_OK", allok)


if __name__ == "__main__":
    main()
