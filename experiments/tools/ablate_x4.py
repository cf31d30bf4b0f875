"""Timing-only ablations of the x4 kernel (variant 3000+mask) on cfg4 (B1 H16 N16384 D128 bf16 non-causal), with the real
shader clock from the per-workgroup trace so the cycles per tile are real cycles.  Needs an EXPERIMENTAL build of the
bf16/128 unit (TFA_LIB=tiny-flash-attention_amd/lib_exp/libtfa_hip.so)."""
import ctypes as C, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
NAMES = {1: "NOEXP", 2: "NODMA", 4: "NOBARRIER", 8: "NOMAX", 16: "NOQK", 32: "NOPV", 64: "NOKREAD", 128: "NOVREAD"}
dev = torch.device("cuda:0")
B, H, N, D = 1, 16, 16384, 128
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
p = ops.make_params(q, k, v, out, lse, False, 1 / math.sqrt(D))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L = _lib.lib()
tiles_per_wg = N // 64
masks = [0, 1, 2, 4, 8, 192, 201, 6, 15, 207]
if len(sys.argv) > 1:
    masks = [int(x) for x in sys.argv[1].split(",")]
base = int(os.environ.get("ABL_BASE", "3000"))
g_, b_, l_ = C.c_int(), C.c_int(), C.c_int()
for rnd in range(2):
    for m in masks:
        _lib.set_variant(base + m)
        ms = C.c_float()
        _lib.check(L.tfa_fwd_time(C.byref(p), 3, 8, s, C.byref(ms)))
        if rnd == 0:
            continue
        _lib.check(L.tfa_fwd_plan(C.byref(p), C.byref(g_), C.byref(b_), C.byref(l_)))
        tb = torch.zeros((g_.value, 8), dtype=torch.int64, device=dev)
        L.tfa_debug_set_trace(C.c_void_p(tb.data_ptr()))
        _lib.check(L.tfa_fwd(C.byref(p), s)); torch.cuda.synchronize()
        L.tfa_debug_set_trace(None)
        t = tb.cpu().numpy()
        loop = (t[:, 2] - t[:, 1]) / tiles_per_wg
        mhz = np.median((t[:, 3] - t[:, 0]) / np.maximum(t[:, 6], 1) * 100.0)
        name = "+".join(n for b, n in NAMES.items() if m & b) or "FULL"
        print(f"mask {m:3d} {name:40s} {ms.value:7.3f} ms  loop {np.median(loop):6.0f} cyc/tile (p10 {np.percentile(loop, 10):6.0f}, p90 {np.percentile(loop, 90):6.0f}), "
              f"prologue {np.median(t[:, 1] - t[:, 0]):6.0f}, epilogue {np.median(t[:, 3] - t[:, 2]):6.0f} cyc, clock {mhz:5.0f} MHz", flush=True)
_lib.set_variant(-1)
