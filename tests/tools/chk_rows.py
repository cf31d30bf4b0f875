import math, sys, torch
sys.path.insert(0, '/root/repo')
from tiny_flash_attention_amd import _lib, ops
from oracle import oracle as O
dev = torch.device('cuda:0')
var = int(sys.argv[1])
for (N, causal) in [(320, False), (384, False), (512, False), (1024, False)]:
    dt, B, H, D = torch.bfloat16, 1, 1, 128
    q, k, v = O.make_inputs(B, H, N, D, dt, seed=3)
    sc = 1 / math.sqrt(D)
    ref = O.tiled_emulation(q, k, v, causal, sc, 64)
    _lib.set_variant(var)
    o32, lse = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc, out_f32=True)
    torch.cuda.synchronize()
    d = (o32.cpu() - ref).abs()[0, 0]          # (N, D)
    rowerr = d.max(dim=1).values
    bad = (rowerr > 2e-3).nonzero().flatten().tolist()
    colerr = d.max(dim=0).values
    badc = (colerr > 2e-3).nonzero().flatten().tolist()
    print(f"v{var} N{N}: bad rows {len(bad)}: {bad[:40]} ... bad cols {len(badc)}: {badc[:40]}")
