"""GQA shapes: time tfa_fwd and report the K/V bytes it has to read at least (each kv head once) per second.
usage: python tools/bench_gqa.py [B,H,Hk,Nq,Nk,D,causal ...]"""
import ctypes as C, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (64, 32, 8, 1, 8192, 128, 1), (16, 32, 8, 1, 32768, 128, 1), (64, 32, 4, 1, 8192, 128, 1), (64, 32, 32, 1, 8192, 128, 1),
    (4, 32, 8, 4096, 4096, 128, 1), (4, 32, 4, 4096, 4096, 128, 1), (4, 32, 32, 4096, 4096, 128, 1), (8, 64, 8, 2048, 2048, 128, 1)]
dev = torch.device("cuda:0")
L = _lib.lib()
for B, H, Hk, Nq, Nk, D, causal in shapes:
    mk = lambda n, h: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    out = torch.empty_like(q); lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, bool(causal), 1 / math.sqrt(D))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl, by, ms = C.c_double(), C.c_double(), C.c_float()
    L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
    t0 = time.time()
    while time.time() - t0 < 0.5:
        _lib.check(L.tfa_fwd_time(C.byref(p), 0, 20, s, C.byref(ms)))
    r = []
    for _ in range(5):
        _lib.check(L.tfa_fwd_time(C.byref(p), 0, 20, s, C.byref(ms))); r.append(ms.value)
    m = sorted(r)[2]
    kvb = 2.0 * B * Hk * Nk * D * 2
    name = _lib.variant_name(L.tfa_fwd_variant(C.byref(p))).split(" ")[0]
    print(f"B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D} causal={causal}: {m * 1e3:9.1f} us = {fl.value / (m * 1e-3) / 1e12:7.1f} TFLOP/s; K+V {kvb / 1e6:8.1f} MB -> {kvb / (m * 1e-3) / 1e9:7.1f} GB/s if read once  [{name}]", flush=True)
