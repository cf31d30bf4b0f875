"""Split-KV measurements: (1) tfa_merge alone against the HBM roofline (algorithmic bytes = nparts fp32 partials read +
16-bit result written), (2) one-pass forward vs split-KV forward on a small-grid shape and on a decode-like shape."""
import ctypes as C, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (P, B, H, N, D) in ((8, 1, 16, 16384, 128), (2, 4, 8, 1024, 64), (8, 8, 32, 1, 128)):
    o = torch.randn((P, B, H, N, D), device=dev)
    l = torch.randn((P, B, H, N), device=dev)
    ms = timeit(lambda: ops.merge_partials(o, l, torch.bfloat16))
    by = o.numel() * 4 + l.numel() * 4 + B * H * N * D * 2 + B * H * N * 4
    print(f"merge P={P} rows={B * H * N} D={D}: {ms * 1e3:8.1f} us  {by / ms / 1e6:8.1f} GB/s ({by / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s), {by / 1e6:.1f} MB")

for (name, B, H, Nq, Nk, D, dt, causal, splits) in (("cfg2", 4, 8, 1024, 1024, 64, torch.float16, False, 2), ("cfg2", 4, 8, 1024, 1024, 64, torch.float16, False, 4),
                                                     ("decode", 8, 32, 1, 16384, 128, torch.bfloat16, True, 8), ("decode", 8, 32, 1, 16384, 128, torch.bfloat16, True, 32),
                                                     ("decode", 1, 32, 1, 16384, 128, torch.bfloat16, True, 16), ("decode", 1, 32, 1, 65536, 128, torch.bfloat16, True, 32),
                                                     ("decode", 1, 8, 16, 32768, 128, torch.bfloat16, True, 32)):
    mk = lambda n: torch.empty((B, H, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
    q, k, v = mk(Nq), mk(Nk), mk(Nk)
    sc = 1 / math.sqrt(D)
    t1 = timeit(lambda: ops.flash_attn_fwd(q, k, v, causal, sc))
    t2 = timeit(lambda: ops.flash_attn_fwd_splitkv(q, k, v, causal, sc, splits=splits, native=False))
    t3 = timeit(lambda: ops.flash_attn_fwd_splitkv(q, k, v, causal, sc, splits=splits, native=True))
    print(f"{name} B{B} H{H} Nq{Nq} Nk{Nk} D{D}: one pass {t1 * 1e3:7.1f} us | x{splits} chunks: {splits} launches + merge {t2 * 1e3:7.1f} us | ONE launch + merge {t3 * 1e3:7.1f} us")
