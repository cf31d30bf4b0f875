"""Quick check of tfa_bwd against the fp64 autograd oracle.  usage: python tests/tools/chk_bwd.py"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
from oracle import oracle as O
dev = torch.device("cuda:0")
cases = [(torch.bfloat16, 1, 2, 2, 256, 256, 128, False), (torch.bfloat16, 1, 2, 2, 256, 256, 128, True),
         (torch.float16, 2, 4, 2, 320, 320, 64, True), (torch.bfloat16, 1, 4, 1, 200, 333, 128, True),
         (torch.float16, 1, 2, 2, 333, 200, 64, False), (torch.bfloat16, 2, 2, 2, 1024, 1024, 128, True),
         (torch.bfloat16, 1, 1, 1, 64, 64, 128, False), (torch.float16, 1, 3, 3, 700, 100, 128, True)]
for (dt, B, H, Hk, Nq, Nk, D, causal) in cases:
    q, k, v = O.make_inputs(B, H, Nq, D, dt, seed=11, Hk=Hk, Nk=Nk)
    g = torch.Generator().manual_seed(5)
    dout = torch.empty((B, H, Nq, D), dtype=torch.float32).normal_(0, 0.5, generator=g).to(dt)
    sc = 1 / math.sqrt(D)
    qd, kd, vd, dod = q.to(dev), k.to(dev), v.to(dev), dout.to(dev)
    out, lse = ops.flash_attn_fwd(qd, kd, vd, causal, sc)
    dq, dk, dv = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, sc, grad_f32=True)
    dq16, dk16, dv16 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, sc)
    torch.cuda.synchronize()
    rq, rk, rv = O.attn_bwd_reference(q, k, v, dout, causal, sc)
    msg = []
    for name, a, a16, r in (("dq", dq, dq16, rq), ("dk", dk, dk16, rk), ("dv", dv, dv16, rv)):
        e = (a.double().cpu() - r).abs().max().item()
        e16 = (a16.double().cpu() - r).abs().max().item()
        msg.append(f"{name}: |ref|max {r.abs().max().item():.3f} err32 {e:.2e} err16 {e16:.2e}")
    print(f"{str(dt)[6:]} B{B} H{H}/{Hk} Nq{Nq} Nk{Nk} D{D} c={int(causal)} | " + " | ".join(msg))
