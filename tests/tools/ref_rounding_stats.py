"""Developer checker (uses the oracle): how far are the product kernels from the REFERENCE's rounding points
(oracle.tiled_emulation = main_torch_only.py:240-260, exact running max) and from their own (tiled_emulation_lazy)?
Prints, per case, the fraction of fp32-output elements outside rtol 1e-3 (+1e-4*A), max |d|/A and the 16-bit ulp histogram."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import ulp16  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tiny_flash_attention_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
for dtype, N, D, causal in ((torch.bfloat16, 2048, 128, True), (torch.bfloat16, 1024, 128, False), (torch.float16, 1024, 64, True),
                            (torch.bfloat16, 4096, 128, True), (torch.float16, 2048, 128, False)):
    q, k, v = O.make_inputs(2, 4, N, D, dtype, seed=31)
    sc = 1.0 / math.sqrt(D)
    out16, _ = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc)
    out32, _ = ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), causal, sc, out_f32=True)
    torch.cuda.synchronize()
    A = O.abs_weighted(q, k, v, causal, sc)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    var = _lib.variant_for(2, 4, 4, N, N, D, causal, _lib.TFA_BF16 if dtype == torch.bfloat16 else _lib.TFA_F16)
    for name, emu in (("reference rounding points", O.tiled_emulation(q, k, v, causal, sc, 64)),
                      ("kernel's own rule", O.tiled_emulation_lazy(q, k, v, causal, sc, 64))):
        d = (out32.cpu() - emu).abs()
        frac = (d > 1e-3 * emu.abs() + 1e-4 * A).float().mean().item()
        worst = (d / (eps * A + 1e-12)).max().item()
        u = ulp16(emu, dtype)
        d16 = (out16.float().cpu() - emu.to(dtype).float()).abs()
        print(f"{dtype} N={N} D={D} causal={causal} variant={_lib.variant_name(var)[:12]} vs {name}: T2 violations {frac:.3e}, "
              f"max|d|/(eps*A) {worst:.3f}, 16-bit: >0ulp {(d16 > 0).float().mean().item():.3e} >1ulp {(d16 > u * (1 + 1e-6)).float().mean().item():.3e} "
              f">2ulp {(d16 > 2 * u).float().mean().item():.3e}", flush=True)
