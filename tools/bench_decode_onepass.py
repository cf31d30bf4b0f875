"""Decode through plain tfa_fwd (one pass; the il kernels' decode instantiations): GPU time per call and K/V TB/s.  usage: python tools/bench_decode_onepass.py"""
import math, os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
dev = torch.device("cuda:0")
for (B, H, Hk, Nq, Nk, D) in ((64, 32, 8, 1, 8192, 128), (8, 32, 8, 1, 8192, 128), (16, 32, 32, 1, 4096, 128), (128, 32, 32, 1, 4096, 128), (32, 16, 16, 1, 8192, 64), (256, 8, 8, 1, 4096, 128)):
    mk = lambda n, h: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    out = torch.empty_like(q); lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, True, 1 / math.sqrt(D))
    ms = C.c_float(); best = 1e9
    for _ in range(3):
        _lib.check(_lib.lib().tfa_fwd_time(C.byref(p), 10, 100, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ms)))
        best = min(best, ms.value)
    kv = 2 * B * Hk * Nk * D * 2
    print(f"B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D} ({kv / 2**20:5.0f} MiB): {best * 1e3:7.1f} us = {kv / best / 1e9:5.2f} TB/s of K/V  [{_lib.variant_name(_lib.lib().tfa_fwd_variant(C.byref(p))).split(' ')[0]}]", flush=True)
