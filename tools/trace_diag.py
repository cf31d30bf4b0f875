"""Per-wave timeline of the iterations a causal pass runs OUTSIDE the hand-scheduled tile loop — the loop's exit, the compiler-scheduled (or generated)
diagonal bodies behind it, the idle iterations — from a -DTFA_IL_TRACEITER build of the traced twin (tfa_fwd_il_pass_prologue.inc: it_stamp).
usage: python tools/trace_diag.py [--lib path/to/libtfa_hip.so] [--pass1] [--wi 0] [--n 2]
Prints, per chosen workgroup and wave, the cycles between consecutive stamps (tag 1 = left the loop at tile jj, 2 = end of an iteration, 3 = behind the loop)."""
import argparse, ctypes as C, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--pass1", action="store_true", help="the light (second) pass of the causal pair (debug flag 128)")
ap.add_argument("--wi", type=int, default=0, help="work item (pair) index: heavy block 15 - wi, light block wi")
ap.add_argument("--n", type=int, default=2, help="workgroups to print")
ap.add_argument("--variant", type=int, default=30)
a = ap.parse_args()
if a.lib:
    os.environ["TFA_LIB"] = os.path.abspath(a.lib)
from tiny_flash_attention_amd import _lib, ops
B, H, N, D = 4, 32, 4096, 128
dev = torch.device("cuda:0")
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
L = _lib.lib()
_lib.set_variant(a.variant)
p = ops.make_params(q, k, v, out, lse, True, 1 / math.sqrt(D))
g, b, l = C.c_int(), C.c_int(), C.c_int()
_lib.check(L.tfa_fwd_plan(C.byref(p), C.byref(g), C.byref(b), C.byref(l)))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    _lib.check(L.tfa_fwd(C.byref(p), s))
buf = torch.zeros((g.value * 8 + 64 * 8 * 16,), dtype=torch.int64, device=dev)
L.tfa_debug_set_flags(128 if a.pass1 else 0)
L.tfa_debug_set_trace(C.c_void_p(buf.data_ptr()))
_lib.check(L.tfa_fwd(C.byref(p), s)); torch.cuda.synchronize()
L.tfa_debug_set_trace(None); L.tfa_debug_set_flags(0); _lib.set_variant(-1)
t = buf.cpu().numpy()
wg = t[: g.value * 8].reshape(g.value, 8)
rec = t[g.value * 8:].view(np.uint64).reshape(64, 8, 16)
wi = wg[:64, 7] & 0xFFFFFFFF
print(f"# {'light' if a.pass1 else 'heavy'} pass of work item {a.wi}; per-workgroup stamps: prologue {np.median(wg[:,1]-wg[:,0]):.0f}, loop {np.median((wg[:,2]-wg[:,1])[(wg[:,7] & 0xFFFFFFFF) == a.wi]):.0f} cycles")
shown = 0
for w in range(64):
    if int(wi[w]) != a.wi or shown >= a.n:
        continue
    shown += 1
    print(f"workgroup {w}:")
    t0 = None
    for wave in range(8):
        r = rec[w, wave]
        cyc = (r & np.uint64(0xFFFFFFFF)).astype(np.int64); tag = (r >> np.uint64(32)).astype(np.int64)
        n = int((r != 0).sum())
        if n == 0:
            print(f"  wave {wave}: no records"); continue
        if t0 is None:
            t0 = cyc[0]
        parts = []
        for i in range(n):
            kind = tag[i] & 0xFF
            d = (cyc[i] - (cyc[i - 1] if i else t0)) & 0xFFFFFFFF
            lab = {0: "start", 1: f"loop->j{tag[i] >> 8}", 2: "it", 3: f"end(nt{(tag[i] >> 8) & 0xFF},nact{(tag[i] >> 16) & 0xFF})"}[int(kind)]
            parts.append(f"{lab}+{d}")
        print(f"  wave {wave}: " + " ".join(parts) + f"   total {(cyc[n - 1] - cyc[0]) & 0xFFFFFFFF}")
