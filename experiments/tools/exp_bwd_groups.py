"""Experiment: the workspace (5-GEMM) backward in GROUPS of heads, so that a group's dS (written by the fused dK/dV launch, read by
the dQ launch right behind it) stays in the 256 MB memory-side cache instead of making a round trip through HBM.
Emulation through the public API: one tfa_bwd call per group of `g` heads of one batch element (views of the same tensors),
sequential on one stream, a ring of two group-sized workspaces.  usage: python tools/exp_bwd_groups.py [--cfg cfg3]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "n2k": (8, 32, 2048, 128, torch.bfloat16, True), "n8k": (2, 16, 8192, 128, torch.bfloat16, True)}
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, H, N, D, dt, causal = CFG[a.cfg]
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
q, k, v, dout = mk(), mk(), mk(), mk()
sc = 1 / math.sqrt(D)
out, lse = ops.flash_attn_fwd(q, k, v, causal, sc)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.empty_like(lse)
L = _lib.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / a.iters)
    return best

p = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
t_def = timed(lambda: _lib.check(L.tfa_bwd(C.byref(p), s)))
ref = (dq.clone(), dk.clone(), dv.clone())
need = ops.bwd_workspace_bytes(p)
ws = torch.empty((need,), dtype=torch.uint8, device=dev)
p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
t_ws = timed(lambda: _lib.check(L.tfa_bwd(C.byref(p), s)))
del ws
print(f"{a.cfg}: default {t_def:.3f} ms | workspace form, one call ({need / 1e9:.2f} GB): {t_ws:.3f} ms")
for g in (2, 4, 8, 16, 32):
    if g > H:
        continue
    plist = []
    for b in range(B):
        for h0 in range(0, H, g):
            sl = lambda t: t[b:b + 1, h0:h0 + g]
            pg = ops.make_bwd_params(sl(q), sl(k), sl(v), sl(out), sl(lse), sl(dout), sl(dq), sl(dk), sl(dv), sl(delta), causal, sc)
            plist.append(pg)
    need_g = ops.bwd_workspace_bytes(plist[0])
    ring = [torch.empty((need_g,), dtype=torch.uint8, device=dev) for _ in range(2)]
    for i, pg in enumerate(plist):
        pg.workspace, pg.workspace_bytes = ring[i & 1].data_ptr(), need_g
    dq.zero_(); dk.zero_(); dv.zero_()
    def run():
        for pg in plist:
            _lib.check(L.tfa_bwd(C.byref(pg), s))
    t = timed(run)
    err = max((x.float() - y.float()).abs().max().item() for x, y in zip((dq, dk, dv), ref))
    print(f"   groups of {g:2d} heads ({len(plist)} calls, dS {need_g / 1e6:.0f} MB per group): {t:.3f} ms   max|d| vs default {err:.2e}")
    del ring
