"""Per-iteration timeline of every wave of a few workgroups (variant 41 = variant 30 + VF_IL_ITERTRACE): for each iteration of
the tile loop the cycle (since the wave's start) before the end-of-tile wait, after the wait and after the barrier.
usage: python tools/trace_iters.py [cfg3|cfg3nc|cfg4] [--wg 0,56] [--variant 41]"""
import argparse, ctypes as C, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False), "cfg4": (1, 16, 16384, 128, torch.bfloat16, False)}
ap = argparse.ArgumentParser()
ap.add_argument("cfg", nargs="?", default="cfg3")
ap.add_argument("--wg", default="0,56")
ap.add_argument("--variant", type=int, default=41)
ap.add_argument("--shape", default="", help="B,H,Nq,Nk,causal(0/1)[,Hk] instead of a named config (bf16, D=128)")
ap.add_argument("--dbg", type=lambda x: int(x, 0), default=0, help="tfa_debug_set_flags value during the traced launches")
ap.add_argument("--full", action="store_true", help="print every iteration (default: the first 3, the last 8 of each pass)")
a = ap.parse_args()
WGS, RECS, NW = 64, 256, 8
B, H, N, D, dt, causal = CFG[a.cfg]
Nk = N
Hk = H
if a.shape:
    sh = [int(x) for x in a.shape.split(",")]
    B, H, N, Nk, c = sh[:5]
    Hk = sh[5] if len(sh) > 5 else H
    causal = bool(c)
dev = torch.device("cuda:0")
mk = lambda h, n: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
q, k, v = mk(H, N), mk(Hk, Nk), mk(Hk, Nk)
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
_lib.set_variant(a.variant)
_lib.debug_set_flags(a.dbg)
p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
L = _lib.lib()
g, b, l = C.c_int(), C.c_int(), C.c_int()
_lib.check(L.tfa_fwd_plan(C.byref(p), C.byref(g), C.byref(b), C.byref(l)))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    _lib.check(L.tfa_fwd(C.byref(p), s))
buf = torch.zeros((g.value * 8 + WGS * NW * RECS * 2,), dtype=torch.int64, device=dev)
L.tfa_debug_set_trace(C.c_void_p(buf.data_ptr()))
_lib.check(L.tfa_fwd(C.byref(p), s)); torch.cuda.synchronize()
L.tfa_debug_set_trace(None)
_lib.set_variant(-1); _lib.debug_set_flags(0)
t = buf.cpu().numpy()
wgrec = t[: g.value * 8].reshape(g.value, 8)
recs = t[g.value * 8:].view(np.uint32).reshape(WGS, NW, RECS, 4)
KIND = {0: "fused", 1: "fusedM", 2: "slow", 3: "tail", 4: "idle", 7: "PASS"}
for wg in [int(x) for x in a.wg.split(",")]:
    wi = int(wgrec[wg, 7] & 0xFFFFFFFF); bh = int(wgrec[wg, 7] >> 32)
    print(f"== workgroup {wg}: head {bh} work item {wi}; life {int(wgrec[wg,3]-wgrec[wg,0])} cycles")
    # per wave: list of records until zero tag/time
    per = []
    for w in range(NW):
        r = recs[wg, w]
        n = int(np.max(np.nonzero(r[:, 1])[0]) + 1) if r[:, 1].any() else 0
        per.append(r[:n])
    npass = 1 + int(max((int(x[-1, 3]) >> 24) for x in per if len(x)))
    for ps in range(npass):
        rows = [x[(x[:, 3] >> 24) == ps] for x in per]
        its = [x[((x[:, 3] >> 16) & 0xFF) != 7] for x in rows]
        summ = [x[((x[:, 3] >> 16) & 0xFF) == 7] for x in rows]
        nit = max(len(x) for x in its)
        print(f"  pass {ps}: {nit} iterations; per wave [prologue end, loop end, own tiles]: " + " ".join(f"w{w}:{int(sx[0,0])},{int(sx[0,1])},{int(sx[0,2])}" for w, sx in enumerate(summ) if len(sx)))
        show = range(nit) if (a.full or nit <= 12) else list(range(3)) + list(range(nit - 8, nit))
        prev = [int(sx[0, 0]) if len(sx) else 0 for sx in summ]
        for i in range(nit):
            cells = []
            for w in range(NW):
                if i < len(its[w]):
                    a0, a1, a2, tag = [int(x) for x in its[w][i]]
                    kind = KIND.get((tag >> 16) & 0xFF, "?")
                    cells.append(f"{kind[:6]:>6s} {a0 - prev[w]:5d}+{a1 - a0:4d}+{a2 - a1:4d}")
                    prev[w] = a2
                else:
                    cells.append(" " * 22)
            if i in show:
                print(f"   it {i:3d} | " + " | ".join(cells))
            elif i == 3:
                print("   ...")
        # steady-state statistics of this pass: iteration period of wave 0 over the fused iterations
        if nit > 12:
            w0 = its[0]
            per_it = np.diff(w0[:, 2].astype(np.int64))
            print(f"   wave 0 period over all iterations: median {np.median(per_it):.0f} mean {per_it.mean():.0f}")
print("(cells: kind, cycles of work since the previous barrier + wait for memory + wait at the barrier)")
