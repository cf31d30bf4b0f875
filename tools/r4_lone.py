"""How fast is ONE wave per SIMD?  256 workgroups (one per CU), one 256-row block each, Nk keys, non-causal: with Nq = 256 all eight waves
work (two per SIMD), with Nq = 128 only waves 0-3 (the decode instantiations skip the tile work of waves without a valid row).
Prints the time per KV tile for each variant: usage: python tools/r4_lone.py --variants 45,46,47 [--nk 8192]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="45,46,47")
ap.add_argument("--nk", type=int, default=8192)
ap.add_argument("--rows", default="256,128,64,32")
ap.add_argument("--hk", type=int, default=8, help="K/V heads (256 query heads): 8 = every K/V head shared by 32 workgroups of one XCD (compute-bound), 256 = private K/V (HBM-bound)")
ap.add_argument("--dbg", type=lambda x: int(x, 0), default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()
H, D = 256, 128
for nq in [int(x) for x in a.rows.split(",")]:
    mk = lambda h, n: torch.empty((1, h, n, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
    q, k, v = mk(H, nq), mk(a.hk, a.nk), mk(a.hk, a.nk)
    out = torch.empty_like(q); lse = torch.empty((1, H, nq), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, False, 1 / math.sqrt(D))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = C.c_float()
    row = []
    for var in [int(x) for x in a.variants.split(",")]:
        _lib.set_variant(var); _lib.debug_set_flags(a.dbg)
        try:
            _lib.check(L.tfa_fwd_time(C.byref(p), 20, 50, s, C.byref(ms)))
            best = 1e9
            for _ in range(5):
                _lib.check(L.tfa_fwd_time(C.byref(p), 2, 30, s, C.byref(ms)))
                best = min(best, ms.value)
        finally:
            _lib.set_variant(-1); _lib.debug_set_flags(0)
        tiles = a.nk // 64
        row.append(f"v{var}: {best*1e3:7.1f} us = {best*1e6/tiles:6.1f} ns/tile")
    print(f"Nq={nq:4d} ({(nq+31)//32} active waves per workgroup), {a.nk//64} tiles: " + "  ".join(row), flush=True)
