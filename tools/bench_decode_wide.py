"""Decode-like shapes at head dims above 128: tfa_fwd (one pass) vs tfa_fwd_splitkv — its one-launch form (the LDS-DMA kernel 256 wide,
chunk index in the grid) and, forced by debug flag 8192, its one-launch-per-chunk route with the chunk launches in line on the
caller's stream (flag 16384) or spread over the side streams.  Wall time per call over 50 back-to-back calls
(host launch costs included).  usage: python tools/bench_decode_wide.py"""
import math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
dev = torch.device("cuda:0")
def wall(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for (B, H, Hk, Nq, Nk, D) in ((1, 8, 8, 1, 16384, 256), (1, 16, 16, 1, 65536, 256), (4, 8, 8, 1, 32768, 192), (1, 8, 2, 4, 32768, 256)):
    mk = lambda n, h: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    sc = 1 / math.sqrt(D)
    ref, _ = ops.flash_attn_fwd(q, k, v, True, sc)
    t1 = wall(lambda: ops.flash_attn_fwd(q, k, v, True, sc))
    res = []
    for splits in (8, 16, 32):
        o, _ = ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=splits)
        err = (o.float() - ref.float()).abs().max().item()
        t0 = wall(lambda: ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=splits))
        res.append(f"{splits} chunks: one launch {t0:7.1f} us = {2 * B * Hk * Nk * D * 2 / t0 / 1e6:5.2f} TB/s (max|d| {err:.1e})")
    _lib.debug_set_flags(8192 | 16384)
    try:
        ta = wall(lambda: ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=4))
        _lib.debug_set_flags(8192)
        tb = wall(lambda: ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=4))
    finally:
        _lib.debug_set_flags(0)
    res.append(f"per-chunk launches (4 chunks) in line {ta:7.1f} us, over side streams {tb:7.1f} us")
    kv_gb = 2 * B * Hk * Nk * D * 2 / 1e9
    print(f"B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D}: one pass {t1:7.1f} us = {kv_gb / t1 * 1e6 / 1e3:5.2f} TB/s of K/V | " + " | ".join(res), flush=True)
