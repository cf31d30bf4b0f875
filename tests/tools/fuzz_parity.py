"""Randomised parity sweep: random (B,H,Hk,Nq,Nk,D,dtype,causal,scale,layout) through chosen kernel variants against the
fp64 oracle (16-bit bar 1e-2 scaled with |O|; rigorous fp32-out bound 2^-8 A / 2^-11 A from rounding P; LSE and its +inf
pattern) and the matching same-rounding-points emulation (rtol 1e-3 outlier fraction, judged only on large cases).
Round 1: 620 cases over variants {auto, 17, 27, 30, 31, 32}: no violation of the rigorous bounds (tests/tools/fuzz_bwd.py: 150 backward cases, none).
usage: python tests/tools/fuzz_parity.py [--cases 150] [--variants -1,27,30] [--seed 0]"""
import argparse, math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=150)
ap.add_argument("--variants", default="-1,27,30")
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
dev = torch.device("cuda:0")
variants = [int(x) for x in a.variants.split(",")]
bad = 0
for case in range(a.cases):
    D = rng.choice([64, 128])
    dt = rng.choice([torch.bfloat16, torch.float16])
    Hk = rng.choice([1, 2, 3])
    H = Hk * rng.choice([1, 2, 4])
    B = rng.choice([1, 2])
    Nq = rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 700, 1000])
    Nk = Nq if rng.random() < 0.5 else rng.choice([1, 5, 63, 64, 65, 130, 256, 333, 500, 640, 777, 1024, 1300])
    causal = rng.random() < 0.6
    scale = rng.choice([1 / math.sqrt(D), 0.02, 0.3, 1.0 / math.sqrt(max(Nq, 2))])
    std = rng.choice([0.5, 1.0, 2.0])
    layout = rng.choice(["bhnd", "bnhd"])
    var = rng.choice(variants)
    q, k, v = O.make_inputs(B, H, Nq, D, dt, seed=1000 + case, std=std, Hk=Hk, Nk=Nk)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    if layout == "bnhd":
        qd, kd, vd = (t.transpose(1, 2).contiguous() for t in (qd, kd, vd))
    _lib.set_variant(var)
    try:
        o16, lse = ops.flash_attn_fwd(qd, kd, vd, causal, scale, layout=layout)
        o32, _ = ops.flash_attn_fwd(qd, kd, vd, causal, scale, layout=layout, out_f32=True)
        torch.cuda.synchronize()
        used = _lib.variant_for(B, H, Hk, Nq, Nk, D, causal)
    finally:
        _lib.set_variant(-1)
    if layout == "bnhd":
        o16, o32 = o16.transpose(1, 2), o32.transpose(1, 2)
    exact, lse_x = O.exact64(q, k, v, causal, scale, return_lse=True)
    A = O.abs_weighted(q, k, v, causal, scale)
    emu = (O.tiled_emulation_lazy if _lib.lazy_reference(var if var >= 0 else used) else O.tiled_emulation)(q, k, v, causal, scale, 64)
    e16 = (o16.float().cpu() - exact).abs().max().item()
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    t3 = ((o32.cpu() - exact).abs() - (eps * A + 1e-6)).max().item()
    viol = ((o32.cpu() - emu).abs() > 1e-3 * emu.abs() + 1e-4 * A).float().mean().item()
    fin = torch.isfinite(lse_x)
    pat = bool((torch.isinf(lse.cpu()) == ~fin).all())
    el = (lse.cpu()[fin] - lse_x[fin]).abs().max().item() if fin.any() else 0.0
    vmax = v.float().abs().max().item()
    # 16-bit bar: the reference's 1e-2 (inputs of std 0.5) scaled with the output magnitude (one ulp16 of the largest |O|);
    # T2 outliers: peaky rows (std 2, scale 0.3) let a single flipped 16-bit rounding of P show, so allow 1e-3 of the elements
    omax = exact.abs().max().item()
    t2_ok = viol <= 1e-3 or o32.numel() < 100000     # few elements: one flipped rounding of P is already > 1e-3 of them
    ok = e16 <= max(1e-2, 2.0 ** -6 * omax) and t3 <= 0 and t2_ok and pat and el <= 1e-4 * max(1.0, lse_x[fin].abs().max().item() if fin.any() else 1.0) and bool(torch.isfinite(o16.float()).all())
    if not ok:
        bad += 1
        print(f"FAIL case {case}: var {var} D{D} {str(dt)[6:]} B{B} H{H}/{Hk} Nq{Nq} Nk{Nk} causal={causal} scale={scale:.4f} std={std} {layout}: "
              f"e16={e16:.2e} t3excess={t3:.2e} t2viol={viol:.2e} lse_pat={pat} lse_err={el:.2e}")
print(f"{a.cases} cases, {bad} failures")
