"""Forward timing of a list of shapes through the C ABI (tfa_fwd_time: HIP events around back-to-back launches).
usage: python tools/bench_shapes.py B,H,N,D,dtype,causal [...]      dtype: f16 | bf16, causal: 0 | 1
With no arguments: the "classic configs" the reference's test script lists (flash_attention_cutlass/test.py:44-48:
batch 4, heads 32/16/8, seqlen 4096, head dim 64/128/256; fp16 causal as the script runs them) and their bf16 / non-causal
siblings.  q, k, v ~ normal(0, 0.5) (test.py:15-19); one second of launches first (clock and power settle)."""
import ctypes as C, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops

INTERLEAVE = "--interleave" in sys.argv     # round-robin over the shapes (7 rounds of 30 launches each) instead of one after the other
shapes = [tuple(a.split(",")) for a in sys.argv[1:] if a != "--interleave"]
if not shapes:
    shapes = [(4, h, 4096, d, dt, c) for (h, d) in ((32, 64), (16, 128), (8, 256)) for dt in ("f16", "bf16") for c in (1, 0)]
dev = torch.device("cuda:0")
L = _lib.lib()
prepared = []
for B, H, N, D, dt, causal in shapes:
    B, H, N, D, causal = int(B), int(H), int(N), int(D), bool(int(causal))
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(tdt)
    q, k, v = mk(), mk(), mk()
    out = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl, by, ms = C.c_double(), C.c_double(), C.c_float()
    L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
    if INTERLEAVE:
        prepared.append((f"B{B} H{H} N{N} D{D} {dt:4s} causal={int(causal)}", p, (q, k, v, out, lse), fl.value, s))
        continue
    t0 = time.time()
    while time.time() - t0 < 1.0:
        _lib.check(L.tfa_fwd_time(C.byref(p), 0, 50, s, C.byref(ms)))
    r = []
    for _ in range(5):
        _lib.check(L.tfa_fwd_time(C.byref(p), 0, 50, s, C.byref(ms)))
        r.append(ms.value)
    m = sorted(r)[2]
    name = _lib.variant_name(L.tfa_fwd_variant(C.byref(p))).split(" ")[0]
    print(f"B{B} H{H} N{N} D{D} {dt:4s} causal={int(causal)}: {m:.3f} ms = {fl.value / (m * 1e-3) / 1e12:7.1f} TFLOP/s = {fl.value / (m * 1e-3) / 2.5e15 * 100:4.1f} % of 2.5 PF  [{name}]", flush=True)

if INTERLEAVE:
    ms = C.c_float()
    t0 = time.time()
    while time.time() - t0 < 1.0:
        _lib.check(L.tfa_fwd_time(C.byref(prepared[0][1]), 0, 50, prepared[0][4], C.byref(ms)))
    res = [[] for _ in prepared]
    for r in range(7):
        for i, (name, p, keep, fl, s) in enumerate(prepared):
            _lib.check(L.tfa_fwd_time(C.byref(p), 3, 30, s, C.byref(ms)))
            res[i].append(ms.value)
    for (name, p, keep, fl, s), r in zip(prepared, res):
        m = sorted(r)[len(r) // 2]
        print(f"{name}: {m:.3f} ms = {fl / (m * 1e-3) / 1e12:7.1f} TFLOP/s (interleaved)", flush=True)
