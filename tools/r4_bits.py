"""Bit-identity of A/B arms against a base variant over a list of awkward shapes (D = 128 arms of round 4: the TAIL / PREF2 kernels
reorder no floating-point operation, so every output bit and every LSE bit must equal variant 30's).
usage: python tools/r4_bits.py --base 30 --arms 38,39,40"""
import argparse, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--base", type=int, default=30)
ap.add_argument("--arms", default="38,39,40")
ap.add_argument("--dims", default="128")
ap.add_argument("--dtypes", default="bf16,f16", help="the round-4 arms of lib_x exist in the bf16 units only: --dtypes bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
arms = [int(x) for x in a.arms.split(",")]
# B, H, Hk, Nq, Nk, causal
SHAPES = [(1, 2, 2, 256, 256, True), (1, 2, 2, 256, 256, False), (2, 3, 3, 512, 512, True), (1, 2, 2, 200, 200, True), (1, 8, 8, 1, 1, True),
          (1, 2, 2, 1024, 1024, True), (1, 4, 2, 128, 384, True), (1, 4, 2, 384, 128, True), (1, 4, 2, 100, 333, False), (1, 4, 2, 1, 1000, True),
          (1, 4, 2, 257, 64, False), (1, 4, 2, 700, 1500, True), (1, 4, 4, 1111, 1111, False), (1, 4, 4, 1111, 1111, True), (2, 8, 8, 2048, 2048, True),
          (1, 16, 16, 4096, 4096, True), (1, 8, 8, 4096, 4096, False), (1, 2, 2, 63, 63, True), (1, 2, 2, 65, 65, True), (1, 2, 2, 320, 320, True),
          (1, 2, 2, 777, 777, True), (1, 2, 1, 513, 513, True), (1, 2, 2, 64, 64, True), (1, 2, 2, 128, 128, True), (1, 2, 2, 192, 192, True)]
bad = 0
for D in [int(x) for x in a.dims.split(",")]:
    for dt in [{"bf16": torch.bfloat16, "f16": torch.float16}[x] for x in a.dtypes.split(",")]:
        for f32 in (False, True):
            for (B, H, Hk, Nq, Nk, causal) in SHAPES:
                g = torch.Generator(device="cpu").manual_seed(B * 7 + H * 13 + Nq + Nk * 3 + D)
                mk = lambda h, n: (torch.randn((B, h, n, D), generator=g) * 0.5).to(dt).to(dev)
                q, k, v = mk(H, Nq), mk(Hk, Nk), mk(Hk, Nk)
                res = {}
                for var in [a.base] + arms:
                    _lib.set_variant(var)
                    try:
                        o, l = ops.flash_attn_fwd(q, k, v, causal, 1 / math.sqrt(D), out_f32=f32)
                        torch.cuda.synchronize()
                        res[var] = (o.clone(), l.clone())
                    finally:
                        _lib.set_variant(-1)
                for var in arms:
                    same = torch.equal(res[var][0].view(torch.int32 if f32 else torch.int16), res[a.base][0].view(torch.int32 if f32 else torch.int16)) and \
                        torch.equal(res[var][1].view(torch.int32), res[a.base][1].view(torch.int32))
                    if not same:
                        bad += 1
                        d = (res[var][0].float() - res[a.base][0].float()).abs().max().item()
                        print(f"DIFF v{var} D{D} {dt} f32out={f32} B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} causal={causal}: max|d|={d:.3e} nan={bool(torch.isnan(res[var][0].float()).any())}", flush=True)
print(f"r4_bits: {'ALL SAME' if bad == 0 else str(bad) + ' DIFFERENCES'} over {len(SHAPES)} shapes x 2 dtypes x 2 output types, arms {arms} vs {a.base}")
sys.exit(1 if bad else 0)
