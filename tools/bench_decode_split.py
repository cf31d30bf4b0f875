"""Decode through tfa_fwd_splitkv (suggested chunk count): wall time per call and K/V TB/s.  usage: python tools/bench_decode_split.py"""
import math, os, sys, time, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
dev = torch.device("cuda:0")
def gpu_ms(fn, n=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for (B, H, Hk, Nq, Nk, D) in ((1, 32, 32, 1, 16384, 128), (1, 32, 32, 1, 65536, 128), (8, 8, 8, 1, 32768, 128), (1, 32, 8, 1, 65536, 128), (4, 16, 16, 1, 32768, 64),
                              (1, 16, 16, 1, 65536, 256), (1, 8, 8, 1, 16384, 256), (4, 8, 8, 1, 32768, 192)):
    mk = lambda n, h: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    sc = 1 / math.sqrt(D)
    out = torch.empty_like(q); lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, True, sc)
    L = _lib.lib()
    splits = int(L.tfa_fwd_suggest_splits(C.byref(p)))
    ms = gpu_ms(lambda: ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=max(splits, 2)))
    print(f"B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D}: {splits:2d} chunks  {ms * 1e3:7.1f} us = {2 * B * Hk * Nk * D * 2 / ms / 1e9:5.2f} TB/s of K/V", flush=True)
