"""Causal pass anatomy from the per-workgroup trace (stamps cover pass 0 = the heavy block; the rest = its epilogue + the
whole light pass): usage: python tools/trace_passes.py VARIANT [cfg3]"""
import ctypes as C, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
variant = int(sys.argv[1]); B, H, N, D = 4, 32, 4096, 128
PASS1 = len(sys.argv) > 2 and sys.argv[2] == "pass1"     # stamps of the light (second) pass: tfa_debug_set_flags(128)
dev = torch.device("cuda:0")
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
_lib.set_variant(variant)
p = ops.make_params(q, k, v, out, lse, True, 1 / math.sqrt(D))
L = _lib.lib()
g, b, l = C.c_int(), C.c_int(), C.c_int()
_lib.check(L.tfa_fwd_plan(C.byref(p), C.byref(g), C.byref(b), C.byref(l)))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    _lib.check(L.tfa_fwd(C.byref(p), s))
buf = torch.zeros((g.value, 8), dtype=torch.int64, device=dev)
L.tfa_debug_set_flags(128 if PASS1 else 0)
L.tfa_debug_set_trace(C.c_void_p(buf.data_ptr()))
_lib.check(L.tfa_fwd(C.byref(p), s)); torch.cuda.synchronize()
L.tfa_debug_set_trace(None)
L.tfa_debug_set_flags(0)
t = buf.cpu().numpy()
wi = t[:, 7] & 0xFFFFFFFF
bm = b.value // 64 * 32 * (2 if b.value == 256 and variant == 33 else 1)
nmb = N // bm
print(f"variant {variant} {_lib.variant_name(variant)[:30]}: block rows {bm}, {nmb} blocks per head, clock {np.median((t[:,3]-t[:,0])/np.maximum(t[:,6],1)*100):.0f} MHz")
for w in np.unique(wi):
    m = wi == w
    nt0 = (bm // 64) * (nmb - w); nt1 = (bm // 64) * (w + 1)
    pro, loop0, rest, life = (t[m, 1] - t[m, 0]), (t[m, 2] - t[m, 1]), (t[m, 3] - t[m, 2]), (t[m, 3] - t[m, 0])
    if PASS1:
        print(f"  item {w}: LIGHT pass ({nt1:3d} tiles): prologue {np.median(pro):6.0f}  loop {np.median(loop0):7.0f} = {np.median(loop0)/max(nt1,1):5.0f}/tile  epilogue {np.median(rest):6.0f}")
        continue
    print(f"  item {w}: heavy {nt0:3d} tiles light {nt1:3d}: prologue {np.median(pro):6.0f}  loop0 {np.median(loop0):7.0f} = {np.median(loop0)/nt0:5.0f}/tile  "
          f"rest(epilogue0 + light pass) {np.median(rest):7.0f} = {np.median(rest) - nt1*np.median(loop0)/nt0:6.0f} beyond {nt1} tiles at the heavy rate;  life {np.median(life):7.0f}")
_lib.set_variant(-1)
