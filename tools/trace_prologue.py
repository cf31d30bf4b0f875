"""Where the first prologue of a workgroup goes (debug build -DTFA_IL_TRACEPRO of the causal bf16 D=128 unit, variant 30): cycles from the wave's
start to (requests out) / (K(0), V(0), K(1) landed) / (Q fragments landed) / (first barrier passed), per launch generation of workgroups.
usage: python tools/trace_prologue.py [--shared] [--nc]      (--shared: every head reads the SAME q/k/v head — stride 0 — so nothing comes from HBM)"""
import ctypes as C, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
B, H, N, D = 4, 32, 4096, 128
shared = "--shared" in sys.argv
causal = "--nc" not in sys.argv
dev = torch.device("cuda:0")
mk = lambda b, h: torch.empty((b, h, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(torch.bfloat16)
if shared:
    q, k, v = (mk(1, 1).expand(B, H, N, D) for _ in range(3))
else:
    q, k, v = mk(B, H), mk(B, H), mk(B, H)
out = torch.empty((B, H, N, D), dtype=torch.bfloat16, device=dev); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
_lib.set_variant(30)
p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
L = _lib.lib()
g, b, l = C.c_int(), C.c_int(), C.c_int()
_lib.check(L.tfa_fwd_plan(C.byref(p), C.byref(g), C.byref(b), C.byref(l)))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    _lib.check(L.tfa_fwd(C.byref(p), s))
buf = torch.zeros((g.value * 8 + g.value * 8 * 4 + g.value * 8 * 8,), dtype=torch.int64, device=dev)
L.tfa_debug_set_trace(C.c_void_p(buf.data_ptr()))
_lib.check(L.tfa_fwd(C.byref(p), s)); torch.cuda.synchronize()
L.tfa_debug_set_trace(None)
_lib.set_variant(-1)
raw = buf.cpu().numpy()
t = raw[: g.value * 8].reshape(g.value, 8).astype(np.uint64)
tw = raw[g.value * 8: g.value * 40].reshape(g.value, 8, 4)
ts = raw[g.value * 40:].reshape(g.value, 8, 8)
iss = (t[:, 4] & np.uint64(0xffffffff)).astype(np.int64); kv = (t[:, 4] >> np.uint64(32)).astype(np.int64); qq = (t[:, 5] >> np.uint64(32)).astype(np.int64)
pro = (t[:, 1] - t[:, 0]).astype(np.int64); life = (t[:, 3] - t[:, 0]).astype(np.int64)
clock = np.median(life / np.maximum(t[:, 6].astype(np.int64), 1) * 100)
print(f"{'shared head (L2-resident inputs)' if shared else 'config 3'}{'' if causal else ' non-causal'}: {g.value} workgroups, clock {clock:.0f} MHz; cycles since the wave's start (wave 0 of each workgroup)")
order = np.argsort(t[:, 0].astype(np.int64), kind="stable")          # by start time: generation g = workgroups 256 g .. 256 g + 255 in start order
pc = lambda x: f"{np.percentile(x,10):6.0f} {np.median(x):6.0f} {np.percentile(x,90):6.0f}"
print(f"  {'generation':>22s} | {'requests out p10 med p90':>26s} | {'K0 V0 K1 landed':>20s} | {'Q landed':>20s} | {'first barrier passed':>20s}")
for gen in range((g.value + 255) // 256):
    m = order[gen * 256:(gen + 1) * 256]
    print(f"  {gen:22d} | {pc(iss[m]):>26s} | {pc(kv[m]):>20s} | {pc(qq[m]):>20s} | {pc(pro[m]):>20s}")

# per wave, relative to the EARLIEST wave start of its workgroup
t0 = tw[:, :, 0].min(axis=1, keepdims=True)
rel = tw - t0[:, :, None]
print("  per wave (median over all workgroups), cycles since the workgroup's first wave started:")
print("   wave | started | requests out | Q landed | at the first barrier")
for w in range(8):
    print(f"   {w:4d} | {np.median(rel[:, w, 0]):7.0f} | {np.median(rel[:, w, 1]):12.0f} | {np.median(rel[:, w, 2]):8.0f} | {np.median(rel[:, w, 3]):8.0f}")
print(f"   last wave at the barrier: median {np.median(rel[:, :, 3].max(axis=1)):.0f}   (which wave is last: {np.bincount(rel[:, :, 3].argmax(axis=1), minlength=8).tolist()})")

if ts[:, :, 0].any():
    e0 = ts[:, :, 0].min(axis=1, keepdims=True)       # the workgroup's first wave entering the kernel
    names = ["kernel entry", "first kernel argument here (t_start)", "work item decoded", "base pointers", "DMA lane offsets", "requests out", "Q landed", "at the first barrier"]
    cols = [ts[:, :, 0], tw[:, :, 0], ts[:, :, 1], ts[:, :, 2], ts[:, :, 3], tw[:, :, 1], tw[:, :, 2], tw[:, :, 3]]
    print("  start-up timeline, median cycles since the workgroup's first wave ENTERED the kernel:")
    print("   " + " | ".join(f"{n[:22]:>22s}" for n in ["wave"] + names))
    for w in range(8):
        print("   " + " | ".join([f"{w:22d}"] + [f"{np.median(c[:, w] - e0[:, 0]):22.0f}" for c in cols]))
