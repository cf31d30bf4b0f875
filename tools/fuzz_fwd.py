"""Randomised cross-check of tfa_fwd (automatic dispatch) against a device fp32 reference: random B, H, Hk, Nq, Nk, D, dtype,
causal, layout.  Bars: |out - ref| <= 1e-2 (the reference's own bar, flash_attention_cutlass/test.py:87), LSE 1e-3, the +inf
pattern of rows that see no key.  usage: python tools/fuzz_fwd.py [--n 300] [--seed 0]"""
import argparse, math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=300)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--decode", action="store_true", help="few query rows against long K/V through the reference-style path (auto_split: tfa_fwd_suggest_splits + tfa_fwd_splitkv)")
ap.add_argument("--spec", action="store_true", help="GQA, a few causal query rows (speculative decoding): the packed-rows path with query positions")
ap.add_argument("--big", action="store_true", help="more heads and batches, longer sequences: grids that fill the chip (il8, paired key-split)")
ap.add_argument("--spike", action="store_true", help="round 6: a few keys aligned with single query rows so that a row's score jumps 2^10 .. 2^160 above its past in one tile "
                                                     "(the max-free rule's power-of-two re-base and its redo; the lazy and exact rules' re-bases)")
a = ap.parse_args()
rng = random.Random(a.seed)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(a.seed)
bad, used = 0, {}
for it in range(a.n):
    D = rng.choice([64, 128, 128, 64, 32, 96, 256, 72, 192, 136, 200, 8, 40, 248])
    dt = rng.choice([torch.bfloat16, torch.float16])
    causal = rng.random() < 0.6
    Hk = rng.choice([1, 2, 4, 8])
    H = Hk * rng.choice([1, 1, 2, 4])
    B = rng.choice([1, 1, 2, 3])
    if a.big:
        Hk = rng.choice([4, 8, 16]); H = Hk * rng.choice([1, 2]); B = rng.choice([1, 2, 4])
    kind = rng.random()
    if kind < 0.35:
        Nq = Nk = rng.choice([64, 128, 256, 512, 1024, 2048, 4096]) + rng.choice([0, 0, 1, -1, 17, -37])
    elif kind < 0.7:
        Nq, Nk = rng.randint(1, 700), rng.randint(1, 3000)
    else:
        Nq, Nk = rng.randint(1, 2500), rng.randint(1, 2500)
    Nq, Nk = max(1, Nq), max(1, Nk)
    if a.big and kind < 0.35:
        Nq = Nk = rng.choice([1024, 2048, 4096, 8192]) + rng.choice([0, 0, 1, -1, 17, -37])
    if a.decode:
        D = rng.choice([64, 128, 96, 32, 256, 192])          # (above 128: one launch per chunk over the side streams)
        B, Hk = rng.choice([1, 1, 2]), rng.choice([1, 2, 4, 8]); H = Hk * rng.choice([1, 2, 4])
        Nq, Nk = rng.choice([1, 1, 2, 7, 16, 33, 128, 200]), rng.randint(4096, 40000)
        layout = "bhnd"
    if a.spec:
        D = rng.choice([64, 128, 96, 32, 72])
        B, Hk = rng.choice([1, 2, 5, 40, 70]), rng.choice([1, 2, 4, 8]); H = Hk * rng.choice([2, 4, 8])     # (large batches: grids beyond one block per CU)
        Nq, Nk = rng.choice([2, 3, 4, 5, 8, 13, 16, 31]), rng.randint(1, 6000 if B < 40 else 5000)
        causal = rng.random() < 0.8
    cap = 4e9 if a.big else 6e8
    if B * H * Nq * Nk > cap:
        Nk = max(1, int(cap / (B * H * Nq)))
    layout = rng.choice(["bhnd", "bnhd"]) if not (a.decode or a.spec) else "bhnd"
    shp = (lambda n, h: (B, h, n, D)) if layout == "bhnd" else (lambda n, h: (B, n, h, D))
    pad = 0 if a.spec else rng.choice([0, 0, 0, 8, 24, 64])               # row stride D + pad: rows that do not follow each other in memory
    def mk(n, h):
        full = torch.empty(shp(n, h)[:3] + (D + pad,), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(dt)
        return full[..., :D] if pad else full
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    mode = "auto" if a.decode else rng.choice(["16", "16", "16", "f32", "chunks", "native", "native-chunks", "exact"])
    sc = rng.choice([1.0 / math.sqrt(D), 0.05, 0.3])
    if a.spike:
        tq, tk = ((lambda t: t) if layout == "bhnd" else (lambda t: t.transpose(1, 2)))(q), ((lambda t: t) if layout == "bhnd" else (lambda t: t.transpose(1, 2)))(k)
        for _ in range(rng.choice([1, 2, 4, 8])):
            b_, hk_ = rng.randrange(B), rng.randrange(Hk)
            row, key = rng.randrange(Nq), rng.randrange(Nk)
            qr = tq[b_, hk_ * (H // Hk), row].float()
            n2 = float((qr * qr).sum()) * sc * 1.4426950408889634
            orders = rng.choice([10.0, 30.0, 45.0, 60.0, 70.0, 100.0, 160.0])          # binary orders the score of (row, key) lands at
            tk[b_, hk_, key] = (qr * (orders / max(n2, 1e-6))).to(dt) if math.isfinite(orders / max(n2, 1e-6)) else tk[b_, hk_, key]
    if mode == "f32":                                    # fp32 debug output
        out, lse = ops.flash_attn_fwd(q, k, v, causal, sc, layout=layout, out_f32=True)
    elif mode == "chunks" and layout == "bhnd" and Nk >= 128 and D <= 128:   # python-driven partial passes + tfa_merge
        out, lse = ops.flash_attn_fwd_splitkv(q, k, v, causal, sc, splits=rng.choice([2, 3, 5]), native=False)
    elif mode in ("native", "native-chunks") and layout == "bhnd" and not pad and Nk >= 128:   # tfa_fwd_splitkv: one launch, or (forced) one per chunk
        if mode == "native-chunks":
            _lib.debug_set_flags(8192)
        try:
            out, lse = ops.flash_attn_fwd_splitkv(q, k, v, causal, sc, splits=rng.choice([2, 3, 5]), native=True)
        finally:
            _lib.debug_set_flags(0)
    elif mode == "exact" and D <= 128:                   # TFA_FWD_EXACT_MAX: the exact-running-max kernel
        out, lse = ops.flash_attn_fwd(q, k, v, causal, sc, layout=layout, exact_max=True)
    else:
        out, lse = ops.flash_attn_fwd(q, k, v, causal, sc, layout=layout, auto_split=a.decode)
    name = _lib.variant_name(_lib.variant_for(B, H, Hk, Nq, Nk, D, causal)).split(" ")[0]
    if a.decode:
        import ctypes as C
        name = "splits=%d" % _lib.lib().tfa_fwd_suggest_splits(C.byref(ops.make_params(q, k, v, out, lse, causal, sc)))
    if mode in ("native", "native-chunks", "exact"):
        name = mode
    used[name] = used.get(name, 0) + 1
    tr = (lambda t: t) if layout == "bhnd" else (lambda t: t.transpose(1, 2))
    qf, kf, vf = tr(q).float(), tr(k).float().repeat_interleave(H // Hk, 1), tr(v).float().repeat_interleave(H // Hk, 1)
    s = torch.matmul(qf, kf.transpose(2, 3)) * sc
    if causal:
        i = torch.arange(Nq, device=dev)[:, None] + (Nk - Nq)
        j = torch.arange(Nk, device=dev)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    lref = torch.logsumexp(s, dim=-1)
    pm = torch.softmax(s, dim=-1).nan_to_num(0.0)
    ref = torch.matmul(pm, vf)
    o = tr(out).float()
    d = (o - ref).abs().max().item()
    fin = torch.isfinite(lref)
    dl = (lse[fin] - lref[fin]).abs().max().item() if bool(fin.any()) else 0.0
    pat = bool((torch.isinf(lse) == ~fin).all())
    # (--spike: rows that one key dominates come out as that key's v row, |o| up to the size of v's entries; a 16-bit output of magnitude 2 is already
    #  7.8e-3 from its neighbour's midpoint and P's own rounding adds 2^-9 of it under every rule but the exact one, which forms P = 1 exactly for the
    #  dominant key — the reference's absolute bar is stated for |o| < 1 and scales with the output there: case 536 of seed 6400, |o| = 2.07, d = 1.05e-2,
    #  inside the rigorous bound 2^-8 * A)
    bar = 1e-2 * max(1.0, ref.abs().max().item()) if a.spike else 1e-2
    ok = bool(torch.isfinite(o).all()) and d <= bar and dl <= 1e-3 and pat
    if not ok:
        bad += 1
        print(f"FAIL #{it} B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D} {dt} causal={causal} {layout} sc={sc:.3f} [{name}]: max|d|={d:.3e} lse {dl:.3e} inf-pattern {pat}", flush=True)
        nf = (~torch.isfinite(o)).nonzero()
        if len(nf):
            print(f"     non-finite: {len(nf)} elements; batches {sorted(set(nf[:, 0].tolist()))} heads {sorted(set(nf[:, 1].tolist()))} rows {sorted(set(nf[:, 2].tolist()))[:10]} cols {sorted(set(nf[:, 3].tolist()))[:16]}", flush=True)
            out2, _ = ops.flash_attn_fwd(q, k, v, causal, sc, layout=layout, auto_split=a.decode)   # again, same inputs: deterministic?
            print(f"     second run of the same call: {int((~torch.isfinite(out2.float())).sum())} non-finite", flush=True)
print(f"{a.n - bad}/{a.n} ok; kernels used: {used}")
sys.exit(1 if bad else 0)
