"""Does one launch of the headline shape run slower than the same work as several smaller launches back to back?
usage: python tools/split_launch_ab.py [--cfg B,H,N,D,causal] [--parts 1,2,4]   (batch split into equal parts, same stream)"""
import argparse, ctypes as C, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="4,32,4096,128,1")
ap.add_argument("--parts", default="1,2,4")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
B, H, N, D, causal = [int(x) for x in a.cfg.split(",")]
dev = torch.device("cuda:0")
L = _lib.lib()
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
sc = 1 / math.sqrt(D)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
arms = {}
for parts in [int(x) for x in a.parts.split(",")]:
    ps = []
    if parts <= B:                                   # split the batch
        step = B // parts
        for i in range(parts):
            sl = slice(i * step, (i + 1) * step)
            ps.append(ops.make_params(q[sl], k[sl], v[sl], out[sl], lse[sl], bool(causal), sc))
    else:                                            # split the heads as well
        hp = parts // B; hs = H // hp
        for b in range(B):
            for j in range(hp):
                hsl = slice(j * hs, (j + 1) * hs)
                ps.append(ops.make_params(q[b:b + 1, hsl], k[b:b + 1, hsl], v[b:b + 1, hsl], out[b:b + 1, hsl], lse[b:b + 1, hsl], bool(causal), sc))
    arms[parts] = ps
def run(ps, n):
    for _ in range(n):
        for p in ps:
            st = L.tfa_fwd(C.byref(p), s)
            assert st == 0, st
t0 = time.time()
while time.time() - t0 < 1.5:
    run(arms[1] if 1 in arms else list(arms.values())[0], 50)
torch.cuda.synchronize()
res = {k_: [] for k_ in arms}
for r in range(a.rounds):
    for parts, ps in arms.items():
        run(ps, 3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(ps, a.iters); e1.record(); e1.synchronize()
        res[parts].append(e0.elapsed_time(e1) / a.iters)
for parts, r in res.items():
    m = sorted(r)[len(r) // 2]
    print(f"{a.cfg}: {parts} launch(es) per pass: {m:.4f} ms = {flops / (m * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
