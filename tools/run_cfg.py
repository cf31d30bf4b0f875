"""Launch one named configuration N times through the library (for profilers: rocprofv3 --pmc around it).
usage: python tools/run_cfg.py [cfg3] [--n 60] [--dbg 0x10000] [--lib path] [--variant -1]"""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("cfg", nargs="?", default="cfg3")
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--dbg", type=lambda x: int(x, 0), default=0)
ap.add_argument("--lib", default="")
ap.add_argument("--variant", type=int, default=-1)
a = ap.parse_args()
if a.lib:
    os.environ["TFA_LIB"] = os.path.abspath(a.lib)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False), "cfg4": (1, 16, 16384, 128, torch.bfloat16, False),
       "cfg5": (8, 32, 4096, 128, torch.bfloat16, True)}
B, H, N, D, dt, causal = CFG[a.cfg]
dev = torch.device("cuda:0")
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
L = _lib.lib()
_lib.set_variant(a.variant); L.tfa_debug_set_flags(a.dbg)
p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = C.c_float()
_lib.check(L.tfa_fwd_time(C.byref(p), 10, a.n, s, C.byref(ms)))
fl, by = C.c_double(), C.c_double()
L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
print(f"{a.cfg} dbg={a.dbg:#x} variant={_lib.variant_name(L.tfa_fwd_variant(C.byref(p)))[:24]}: {ms.value:.4f} ms = {fl.value / ms.value / 1e9:.1f} TF")
L.tfa_debug_set_flags(0); _lib.set_variant(-1)
