"""Where do the fused dK/dV kernel's waves wait?  Needs a library built with -DTFA_BWD_TRACE (see experiments/README.md):
   TFA_LIB=.../lib_tr/libtfa_hip.so python tools/trace_bwd_kv.py [--cfg cfg3]
Per role (0: S->P->dV, 1: dP->dS->dK): share of the wave's life spent in the end-of-tile `s_waitcnt vmcnt(0) lgkmcnt(0)` (memory)
and in the `s_barrier` behind it (the other role / the DMA waves)."""
import argparse, ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops
CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False), "cfg4": (1, 16, 16384, 128, torch.bfloat16, False)}
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--ws", action="store_true", help="the workspace (5-GEMM) form: the same launch also stores dS")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, H, N, D, dt, causal = CFG[a.cfg]
mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt)
q, k, v, dout = mk(), mk(), mk(), mk()
sc = 1 / math.sqrt(D)
out, lse = ops.flash_attn_fwd(q, k, v, causal, sc)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.empty_like(lse)
p = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
need = ops.bwd_workspace_bytes(p) if a.ws else 0
ws = torch.zeros((need + (64 << 20),), dtype=torch.uint8, device=dev)          # the trace lives in the last 64 MiB
p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
_lib.debug_bwd_split(4)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    _lib.check(_lib.lib().tfa_bwd(C.byref(p), s))
torch.cuda.synchronize()
_lib.debug_bwd_split(0)
t = ws[need:].view(torch.int64).cpu().view(-1, 4)
nwg = (N // 128) * B * H if True else 0
t = t[: nwg * 8].view(nwg, 8, 4).double()
tot, mem, bar = t[..., 0], t[..., 1], t[..., 2]
t3 = t[..., 3].long()
nu, pro, epi = (t3 & 0xffff).double(), ((t3 >> 16) & 0xffffff).double(), ((t3 >> 40) & 0xffffff).double()
# workgroup anatomy: kernel entry -> tile loop (max over waves), the loop, behind the loop (stores issued; max over waves)
wg_pro, wg_loop, wg_epi, wg_nu = pro.max(dim=1).values, tot.max(dim=1).values, epi.max(dim=1).values, nu[:, 0]
print(f"workgroup anatomy (cycles): entry->loop median {wg_pro.median():.0f} (p10 {wg_pro.quantile(0.1):.0f} p90 {wg_pro.quantile(0.9):.0f}); behind the loop median {wg_epi.median():.0f}; "
      f"loop per tile median {(wg_loop / wg_nu.clamp(min=1)).median():.0f}; sum over workgroups: entry->loop {100 * wg_pro.sum() / (wg_pro + wg_loop + wg_epi).sum():.1f} %, "
      f"loop {100 * wg_loop.sum() / (wg_pro + wg_loop + wg_epi).sum():.1f} %, behind {100 * wg_epi.sum() / (wg_pro + wg_loop + wg_epi).sum():.1f} %")
for lo, hi_ in ((1, 8), (8, 24), (24, 48), (48, 65)):
    m = (wg_nu >= lo) & (wg_nu < hi_)
    if m.any():
        print(f"   workgroups with {lo}..{hi_ - 1} tiles: {int(m.sum())}; entry->loop {wg_pro[m].median():.0f}, loop per tile {(wg_loop[m] / wg_nu[m]).median():.0f}, behind the loop {wg_epi[m].median():.0f}")
print(f"{a.cfg}{' (workspace form)' if a.ws else ''}: {nwg} workgroups x 8 waves; wave life mean {tot.mean():.0f} cycles (100 MHz ticks x?), tiles per workgroup {nu.min():.0f}..{nu.max():.0f}")
for role, sl in (("role 0 (waves 0-3: S, P, dV)", slice(0, 4)), ("role 1 (waves 4-7: dP, dS, dK)", slice(4, 8))):
    T, M, Bq = tot[:, sl].sum(), mem[:, sl].sum(), bar[:, sl].sum()
    print(f"  {role}: memory wait {100 * M / T:5.1f} %   barrier wait {100 * Bq / T:5.1f} %   per tile: life {T / nu[:, sl].sum():.1f}, memory {M / nu[:, sl].sum():.1f}, barrier {Bq / nu[:, sl].sum():.1f} ticks")
for w in range(8):
    print(f"    wave {w}: memory {100 * mem[:, w].sum() / tot[:, w].sum():5.1f} %  barrier {100 * bar[:, w].sum() / tot[:, w].sum():5.1f} %")
