import math, os, sys, ctypes as C
import torch
sys.path.insert(0, "/root/repo")
from tiny_flash_attention_amd import _lib, ops
dev = torch.device("cuda:0")
def gpu_ms(fn, n=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for (B, H, Hk, Nq, Nk, D) in ((12, 32, 8, 1, 16384, 128), (16, 32, 8, 1, 16384, 128), (20, 32, 8, 1, 16384, 128), (3, 32, 32, 1, 16384, 128), (4, 32, 32, 1, 16384, 128), (6, 32, 32, 1, 8192, 128), (24, 8, 8, 1, 32768, 64)):
    mk = lambda n, h: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    sc = 1 / math.sqrt(D)
    t1 = gpu_ms(lambda: ops.flash_attn_fwd(q, k, v, True, sc))
    res = [f"one pass {t1*1e3:6.1f} us"]
    for s in (2, 3, 4):
        t = gpu_ms(lambda: ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=s))
        res.append(f"{s} chunks {t*1e3:6.1f}")
    out = torch.empty_like(q); lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, True, sc)
    print(f"B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D}: blocks {B*Hk:3d}, suggested {_lib.lib().tfa_fwd_suggest_splits(C.byref(p))} | " + " | ".join(res), flush=True)
