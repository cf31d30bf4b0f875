"""Per-workgroup cycle trace (tfa_debug_set_trace): prologue / main loop / epilogue cost and the
occupancy timeline of one launch.  usage: python tools/trace_wg.py [variant] [cfg] """
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops  # noqa: E402

CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False), "cfg2": (4, 8, 1024, 64, torch.float16, False),
       "d256c": (4, 8, 4096, 256, torch.bfloat16, True), "d256": (4, 8, 4096, 256, torch.bfloat16, False)}


def trace(variant, cfg):
    B, H, N, D, dt, causal = CFG[cfg]
    dev = torch.device("cuda:0")
    mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
    q, k, v = mk(), mk(), mk()
    out = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    _lib.set_variant(variant)
    p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
    g, b, l = C.c_int(), C.c_int(), C.c_int()
    _lib.check(_lib.lib().tfa_fwd_plan(C.byref(p), C.byref(g), C.byref(b), C.byref(l)))
    buf = torch.zeros((g.value, 8), dtype=torch.int64, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.check(_lib.lib().tfa_fwd(C.byref(p), s))
    torch.cuda.synchronize()
    _lib.lib().tfa_debug_set_trace(C.c_void_p(buf.data_ptr()))
    _lib.check(_lib.lib().tfa_fwd(C.byref(p), s))
    torch.cuda.synchronize()
    _lib.lib().tfa_debug_set_trace(None)
    t = buf.cpu().numpy().astype(np.int64)
    xcc = t[:, 5] & 0xF
    hw = (t[:, 5] >> 32) & 0xFFFFFFFF
    mhz = (t[:, 3] - t[:, 0]) / np.maximum(t[:, 6], 1) * 100.0
    print(f"shader clock while the workgroups ran (s_memtime / s_memrealtime): median {np.median(mhz):.0f} MHz, p5 {np.percentile(mhz, 5):.0f}, p95 {np.percentile(mhz, 95):.0f}")
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)   # cu_id | sh_id | se_id
    cuid = xcc * 256 + cu
    # s_memtime is per-XCC: normalise each XCC to its own first start
    # s_memtime is not synchronised across CUs/SEs: normalise each CU to its own first start
    # (every CU receives its first workgroup within ~1 us of the launch)
    st = np.zeros(len(t), dtype=np.int64)
    en = np.zeros(len(t), dtype=np.int64)
    for c in np.unique(cuid):
        m = cuid == c
        b0 = t[m, 0].min()
        st[m] = t[m, 0] - b0
        en[m] = t[m, 3] - b0
    pro, loop, epi, nt = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] & 0xFFFFFFFF
    nslow = (t[:, 4] >> 32) & 0xFFFF
    ntrig = (t[:, 4] >> 48) & 0xFFFF
    if nslow.max() > 0:
        print(f"slow-path tiles of wave 0 per workgroup: mean {nslow.mean():.2f} max {nslow.max()} (of {nt.mean():.1f} tiles); of those re-base triggers: mean {ntrig.mean():.2f} max {ntrig.max()}")
        if os.environ.get("DUMP_SLOW"):
            wi_ = t[:, 7] & 0xFFFFFFFF
            for w in np.unique(wi_):
                m = wi_ == w
                print(f"    work item {w}: tiles {nt[m].mean():.0f} slow {nslow[m].mean():.1f} trig {ntrig[m].mean():.1f}")
    span = en.max()
    print(f"== variant {variant} {_lib.variant_name(variant)} | {cfg}: grid {g.value} block {b.value}")
    print(f"kernel span {span} cycles (max over XCCs of last end - first start); prologue mean {pro.mean():.0f} "
          f"(max {pro.max()}), epilogue mean {epi.mean():.0f} (max {epi.max()})")
    A = np.stack([nt, np.ones_like(nt)], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, loop.astype(np.float64), rcond=None)
    print(f"main loop: {coef[0]:.0f} cycles per KV tile + {coef[1]:.0f} fixed; mean tiles {nt.mean():.1f}; mean WG lifetime {(en - st).mean():.0f}")
    ucu = np.unique(cuid)
    wg_per_cu = 2 if b.value <= 256 else 1
    busy = (en - st).sum()
    print(f"distinct CUs seen: {len(ucu)}; occupancy of WG slots = {busy / (len(ucu) * wg_per_cu * span):.3f}")
    ends, gaps, nwg = [], [], []
    for c in ucu:
        m = cuid == c
        o = np.argsort(st[m])
        s_, e_ = st[m][o], en[m][o]
        ends.append(e_.max())
        nwg.append(m.sum())
        if wg_per_cu == 1 and len(s_) > 1:
            gaps.extend((s_[1:] - e_[:-1]).tolist())
    ends = np.array(ends)
    print(f"per-CU finish time: min {ends.min()} p50 {int(np.median(ends))} max {ends.max()}  (ideal = mean busy {busy / (len(ucu) * wg_per_cu):.0f}); WGs per CU min/max {min(nwg)}/{max(nwg)}")
    if gaps:
        gaps = np.array(gaps)
        print(f"gap between a WG's end and the next WG's start on the same CU: mean {gaps.mean():.0f} p50 {np.median(gaps):.0f} p95 {np.percentile(gaps,95):.0f} max {gaps.max()}  (n={len(gaps)})")
    for x in range(8):
        m = xcc == x
        if m.any():
            bhs = (t[m, 7] >> 32)
            print(f"  XCC{x}: WGs {m.sum()} heads {len(np.unique(bhs))} (bh%8 in {sorted(set((bhs % 8).tolist()))}) first start {st[m].min()} last end {en[m].max()} mean first-16 start {np.sort(st[m])[:32].mean():.0f}")
    # real time (s_memrealtime, 100 MHz): per XCC the shader clock its workgroups saw and the busy time of its CUs — unequal XCC
    # clocks under a shared power budget make the slowest XCC the launch's critical path (in-order round-robin dispatch gives
    # every XCC the same number of workgroups)
    rt_us = t[:, 6] / 100.0
    per_x = []
    for x in range(8):
        m = xcc == x
        if m.any():
            cu_busy = np.array([rt_us[m & (cuid == c)].sum() for c in np.unique(cuid[m])])
            per_x.append((x, float(np.median(mhz[m])), float(cu_busy.mean()), float(cu_busy.max())))
    if per_x:
        print("  real time per XCC: " + "  ".join(f"X{x}: {mz:.0f} MHz, CU busy {bm:.1f} us (max {bx:.1f})" for x, mz, bm, bx in per_x))
        bm = np.array([a[2] for a in per_x])
        print(f"  XCC busy-time spread: slowest / mean = {bm.max() / bm.mean():.3f}, slowest / fastest = {bm.max() / bm.min():.3f}")
    if os.environ.get("DUMP_CU"):
        for c in ucu[:: max(1, len(ucu) // 4)][:4]:
            m = np.where(cuid == c)[0]
            m = m[np.argsort(st[m])]
            print(f"  CU {c:5d}:", " ".join(f"[id{i} nt{nt[i]} {st[i]}..{en[i]}]" for i in m))
    o = np.argsort(st)
    print("start-time quantiles:", [int(st[o[int(q * (len(o) - 1))]]) for q in (0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0)])


if __name__ == "__main__":
    variants = [int(sys.argv[1])] if len(sys.argv) > 1 else [1, 2]
    cfgs = sys.argv[2:] if len(sys.argv) > 2 else ["cfg3", "cfg3nc"]
    for c in cfgs:
        for vv in variants:
            trace(vv, c)
