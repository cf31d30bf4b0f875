"""Timing-only ablations (variant 100+mask) of the dma8 kernel on cfg4 (non-causal, N=16384)."""
import ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
NAMES = {1: "NOQK", 2: "NOPV", 8: "NOSM", 16: "NOKREAD", 32: "NOVREAD", 64: "NODMA", 256: "NOBARRIER"}
dev = torch.device("cuda:0")
B, H, N, D = 1, 16, 16384, 128
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
p = ops.make_params(q, k, v, out, lse, False, 1 / math.sqrt(D))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
masks = [0, 1, 2, 3, 8, 16, 32, 48, 64, 3 + 8, 48 + 8, 48 + 64, 3 + 48 + 64, 8 + 48 + 64, 256, 256 + 64, 256 + 8 + 48 + 64, 3 + 8 + 48]
tiles = B * H * (N // 256) * (N // 64)
for rnd in range(2):
    for m in masks:
        _lib.set_variant(100 + m)
        ms = C.c_float()
        _lib.check(_lib.lib().tfa_fwd_time(C.byref(p), 2, 5, s, C.byref(ms)))
        name = "+".join(n for b, n in NAMES.items() if m & b) or "FULL"
        # cycles per (CU, tile) at 2.1 GHz nominal: 1024 WGs over 256 CUs -> 4 WGs per CU sequentially
        cyc = ms.value * 1e-3 * 2.1e9 / (tiles / 256)
        if rnd == 1:
            print(f"mask {m:3d} {name:28s} {ms.value:7.3f} ms  ~{cyc:6.0f} cyc/tile/CU @2.1GHz")
_lib.set_variant(-1)
