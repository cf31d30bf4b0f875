"""Randomised cross-check of tfa_bwd against fp32 autograd on the device: random B, H, Hk, Nq, Nk, D (multiples of 8 up to
256), dtype, causal, layout.  Bar per gradient: max|d| <= 2e-2 * max|ref| + 1e-3 (16-bit P, dS and outputs; the per-element
bounds are tests/test_bwd_gpu.py's job).  usage: python tools/fuzz_bwd.py [--n 150] [--seed 0] [--focus asm]
--focus asm: the shapes that run the hand-scheduled statements of both launches (128 wide, the default form): several whole tiles per block, lengths that are
and are not multiples of 64 / 128 / 256, Nq != Nk under the bottom-right causal mask, GQA / MQA, both 16-bit types, both layouts."""
import argparse, math, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--focus", default="", choices=["", "asm"])
a = ap.parse_args()
rng = random.Random(a.seed)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(a.seed)
bad = 0
for it in range(a.n):
    D = rng.choice([64, 128, 128, 64, 32, 96, 72, 8, 120, 160, 256, 200])
    dt = rng.choice([torch.bfloat16, torch.float16])
    causal = rng.random() < 0.6
    Hk = rng.choice([1, 2, 4])
    H = Hk * rng.choice([1, 1, 2, 4])
    B = rng.choice([1, 1, 2])
    if rng.random() < 0.4:
        Nq = Nk = max(1, rng.choice([64, 128, 256, 512, 1024]) + rng.choice([0, 0, 1, -1, 17, -37]))
    else:
        Nq, Nk = rng.randint(1, 900), rng.randint(1, 1500)
    if a.focus == "asm":
        D = rng.choice([128, 128, 128, 104, 120])
        Hk = rng.choice([1, 2, 3])
        H = Hk * rng.choice([1, 2, 4])
        B = rng.choice([1, 2])
        base = rng.choice([320, 512, 640, 768, 1024, 1280, 1536, 2048])
        Nq = base + rng.choice([0, 0, 0, 64, -64, 128, 1, -1, 37])
        Nk = Nq if rng.random() < 0.5 else max(64, Nq + rng.choice([-256, -64, 64, 192, 448, 1, -37]))
    layout = rng.choice(["bhnd", "bnhd"])
    shp = (lambda n, h: (B, h, n, D)) if layout == "bhnd" else (lambda n, h: (B, n, h, D))
    mk = lambda n, h: torch.empty(shp(n, h), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(dt)
    q, k, v, dout = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk), mk(Nq, H)
    sc = rng.choice([1.0 / math.sqrt(D), 0.05, 0.2])
    out, lse = ops.flash_attn_fwd(q, k, v, causal, sc, layout=layout)
    mode = rng.choice(["default", "default", "workspace", "split", "windowed"])     # tfa_bwd's three forms (tests/test_bwd_gpu.py: MODES) + the >= 2 GiB instantiations, forced
    if mode == "windowed" and D > 128:
        mode = "default"
    if a.focus == "asm" and rng.random() < 0.8:
        mode = "default"
    f32 = rng.random() < 0.25                          # fp32 gradients
    _lib.debug_bwd_split({"split": 1, "windowed": 2}.get(mode, 0))
    try:
        dq, dk, dv = ops.flash_attn_bwd(q, k, v, out, lse, dout, causal, sc, layout=layout, grad_f32=f32, workspace=True if mode == "workspace" else None)
    finally:
        _lib.debug_bwd_split(False)
    tr = (lambda t: t) if layout == "bhnd" else (lambda t: t.transpose(1, 2))
    qf, kf, vf = (tr(t).float().detach().requires_grad_(True) for t in (q, k, v))
    ke, ve = kf.repeat_interleave(H // Hk, 1), vf.repeat_interleave(H // Hk, 1)
    s = torch.matmul(qf, ke.transpose(2, 3)) * sc
    if causal:
        i = torch.arange(Nq, device=dev)[:, None] + (Nk - Nq)
        j = torch.arange(Nk, device=dev)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    pm = torch.softmax(s, dim=-1).nan_to_num(0.0)
    ref = torch.matmul(pm, ve)
    ref.backward(tr(dout).float())
    msg = []
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        gt = tr(got).float()
        d = (gt - want).abs().max().item()
        bar = 2e-2 * want.abs().max().item() + 1e-3
        if not (bool(torch.isfinite(gt).all()) and d <= bar):
            msg.append(f"{name} max|d|={d:.3e} > {bar:.3e}")
    if msg:
        bad += 1
        print(f"FAIL B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D} {dt} causal={causal} {layout} {mode}{' f32' if f32 else ''} sc={sc:.3f}: " + "; ".join(msg), flush=True)
print(f"{a.n - bad}/{a.n} ok")
sys.exit(1 if bad else 0)
