import math, sys, torch
sys.path.insert(0, '/root/repo')
from tiny_flash_attention_amd import _lib, ops
from oracle import oracle as O
dev = torch.device('cuda:0')
shapes = [(torch.bfloat16, 1, 2, 712, 128, True, None, None), (torch.float16, 1, 4, 300, 64, False, 2, 450),
          (torch.bfloat16, 1, 2, 512, 128, False, None, None), (torch.float16, 2, 2, 520, 64, True, None, None),
          (torch.bfloat16, 1, 2, 2048, 128, False, None, None), (torch.bfloat16, 1, 2, 2048, 128, True, None, None),
          (torch.bfloat16, 1, 1, 64, 128, False, None, None), (torch.bfloat16, 1, 1, 192, 128, False, None, None),
          (torch.bfloat16, 1, 1, 256, 128, True, None, None)]
for var in [19] + [int(x) for x in sys.argv[1].split(',')]:
    for (dt, B, H, N, D, causal, Hk, Nk) in shapes:
        q, k, v = O.make_inputs(B, H, N, D, dt, seed=3, Hk=Hk, Nk=Nk)
        sc = 1 / math.sqrt(D)
        emu = O.tiled_emulation_lazy if 'il' in _lib.variant_name(var)[:3] else O.tiled_emulation
        ref, lref = emu(q, k, v, causal, sc, 64, return_lse=True)
        _lib.set_variant(var)
        errs = []
        for rep in range(3):
            o32, lse = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc, out_f32=True)
            torch.cuda.synchronize()
            d = (o32.cpu() - ref).abs()
            errs.append(d.max().item())
        el = (lse.cpu() - lref).abs().max().item()
        idx = (d == d.max()).nonzero()[0].tolist()
        print(f"v{var} {str(dt)[6:]} N{N} Nk{Nk} D{D} c={int(causal)}: e32 max over 3 runs {max(errs):.2e} (min {min(errs):.2e}) lse {el:.2e} at {idx}  mean {d.mean().item():.2e}")
