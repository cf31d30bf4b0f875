"""Randomised sweep of the backward pass: random (B,H,Hk,Nq,Nk,D,dtype,causal,scale,layout) against the fp64 autograd
oracle with the rigorous 16-bit-rounding bounds of tests/test_bwd_gpu.py (B1 on fp32 gradients, B3 on 16-bit ones).
usage: python tests/tools/fuzz_bwd.py [--cases 100] [--seed 0]"""
import argparse, math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import ops
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
dev = torch.device("cuda:0")
bad = 0
for case in range(a.cases):
    D = rng.choice([64, 128])
    dt = rng.choice([torch.bfloat16, torch.float16])
    Hk = rng.choice([1, 2, 3])
    H = Hk * rng.choice([1, 2, 4])
    B = rng.choice([1, 2])
    Nq = rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 700])
    Nk = Nq if rng.random() < 0.5 else rng.choice([1, 5, 63, 64, 65, 130, 256, 257, 333, 500, 640, 777])
    causal = rng.random() < 0.6
    scale = rng.choice([1 / math.sqrt(D), 0.02, 0.3])
    std = rng.choice([0.5, 1.0, 2.0])
    layout = rng.choice(["bhnd", "bnhd"])
    q, k, v = O.make_inputs(B, H, Nq, D, dt, seed=2000 + case, std=std, Hk=Hk, Nk=Nk)
    dout = torch.empty((B, H, Nq, D)).normal_(0, 0.5, generator=torch.Generator().manual_seed(3000 + case)).to(dt)
    qd, kd, vd, dod = q.to(dev), k.to(dev), v.to(dev), dout.to(dev)
    if layout == "bnhd":
        qd, kd, vd, dod = (t.transpose(1, 2).contiguous() for t in (qd, kd, vd, dod))
    out, lse = ops.flash_attn_fwd(qd, kd, vd, causal, scale, layout=layout)
    g32 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, scale, layout=layout, grad_f32=True)
    g16 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, scale, layout=layout)
    torch.cuda.synchronize()
    if layout == "bnhd":
        g32 = tuple(t.transpose(1, 2) for t in g32); g16 = tuple(t.transpose(1, 2) for t in g16); out = out.transpose(1, 2)
    ref = O.attn_bwd_reference(q, k, v, dout, causal, scale)
    bounds = O.attn_bwd_bounds(q, k, v, out.cpu(), dout, causal, scale)
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    msgs = []
    for name, a32, a16, r, A in zip(("dq", "dk", "dv"), g32, g16, ref, bounds):
        ex = ((a32.double().cpu() - r).abs() - (eps * A + 1e-6)).max().item()
        e16 = (a16.double().cpu() - r).abs().max().item()
        if ex > 0 or not bool(torch.isfinite(a16.float()).all()) or e16 > 1e-2 * max(1.0, r.abs().max().item()):
            msgs.append(f"{name}: excess {ex:.2e} e16 {e16:.2e} |ref| {r.abs().max().item():.2e}")
    if msgs:
        bad += 1
        print(f"FAIL case {case}: D{D} {str(dt)[6:]} B{B} H{H}/{Hk} Nq{Nq} Nk{Nk} causal={causal} scale={scale:.4f} std={std} {layout}: " + "; ".join(msgs))
print(f"{a.cases} cases, {bad} failures")
