"""oracle/oracle.py — Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (tiny-flash-attention_amd/) never does.

Three layers, all CPU:
  * C restatement (oracle/ref_attn.c via ctypes): ``naive``, ``online`` (fp32, following
    flash_attention_c/csrc/attn.cpp:35-98 and :101-169) and ``exact64`` (fp64 ground truth of
    the GPU contract incl. 16-bit P rounding, flash_attention.cu:263-316,601,608-630).
  * ``tiled_emulation``: restatement of the reference's torch tile loop
    (flash_attention_py/main_torch_only.py:160-270): block_m x block_n tiles, causal mask,
    scale, P cast to the input dtype before PV.
  * ``ref_kernels()``: the reference's OWN compiled CPU path (oracle/_ref/_kernels*.so, built
    unmodified from /root/reference by oracle/Makefile) when present.
"""
import ctypes as C
import glob
import importlib.util
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


class _Args(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("Hk", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("D", C.c_int),
        ("qs", C.c_int64 * 3), ("ks", C.c_int64 * 3), ("vs", C.c_int64 * 3), ("os", C.c_int64 * 3),
        ("scale", C.c_float), ("causal", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        for name in ("oracle_attn_naive", "oracle_attn_online"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.POINTER(_Args)]
        L.oracle_attn_exact64.restype = C.c_int
        L.oracle_attn_exact64.argtypes = [C.POINTER(_Args), C.c_int]
        L.oracle_round_bf16.restype = C.c_float
        L.oracle_round_bf16.argtypes = [C.c_float]
        L.oracle_round_fp16.restype = C.c_float
        L.oracle_round_fp16.argtypes = [C.c_float]
        L.oracle_num_threads.restype = C.c_int
        _lib = L
    return _lib


def num_threads():
    return lib().oracle_num_threads()


_P_ROUND = {None: 0, "none": 0, torch.float32: 0, torch.bfloat16: 1, "bf16": 1, torch.float16: 2, "fp16": 2}


def _run(kind, q, k, v, is_causal, softmax_scale, p_round=None, return_lse=False):
    """q (B,H,Nq,D), k/v (B,Hk,Nk,D): CPU tensors of any float dtype (upcast to fp32 exactly)."""
    q32 = q.detach().to("cpu", torch.float32).contiguous()
    k32 = k.detach().to("cpu", torch.float32).contiguous()
    v32 = v.detach().to("cpu", torch.float32).contiguous()
    B, H, Nq, D = q32.shape
    _, Hk, Nk, _ = k32.shape
    out = torch.empty_like(q32)
    lse = torch.empty((B, H, Nq), dtype=torch.float32)
    a = _Args()
    a.q, a.k, a.v, a.o, a.lse = q32.data_ptr(), k32.data_ptr(), v32.data_ptr(), out.data_ptr(), lse.data_ptr()
    a.B, a.H, a.Hk, a.Nq, a.Nk, a.D = B, H, Hk, Nq, Nk, D
    for name, t in (("qs", q32), ("ks", k32), ("vs", v32), ("os", out)):
        arr = getattr(a, name)
        arr[0], arr[1], arr[2] = t.stride(0), t.stride(1), t.stride(2)
    a.scale = float(softmax_scale)
    a.causal = 1 if is_causal else 0
    if kind == "naive":
        rc = lib().oracle_attn_naive(C.byref(a))
    elif kind == "online":
        rc = lib().oracle_attn_online(C.byref(a))
    else:
        rc = lib().oracle_attn_exact64(C.byref(a), _P_ROUND[p_round])
    assert rc == 0
    return (out, lse) if return_lse else out


def naive_attn(q, k, v, is_causal, softmax_scale, return_lse=False):
    """C restatement of run_naive_attn (flash_attention_c/csrc/attn.cpp:35-98)."""
    return _run("naive", q, k, v, is_causal, softmax_scale, return_lse=return_lse)


def flash_attn(q, k, v, is_causal, softmax_scale, return_lse=False):
    """C restatement of run_flash_attn (flash_attention_c/csrc/attn.cpp:101-169)."""
    return _run("online", q, k, v, is_causal, softmax_scale, return_lse=return_lse)


def exact64(q, k, v, is_causal, softmax_scale, p_round=None, return_lse=False):
    """fp64 ground truth; ``p_round`` in {None, torch.bfloat16, torch.float16} rounds P before PV."""
    return _run("exact64", q, k, v, is_causal, softmax_scale, p_round=p_round, return_lse=return_lse)


def sdpa_reference(q, k, v, is_causal, softmax_scale):
    """torch fp32 reference of the same op (the comparison the reference's scripts use,
    flash_attention_c/test.py:9-19, flash_attention_cutlass/test.py:19-27): softmax in fp32."""
    q32, k32, v32 = (t.detach().to("cpu", torch.float32) for t in (q, k, v))
    if k32.shape[1] != q32.shape[1]:
        rep = q32.shape[1] // k32.shape[1]
        k32 = k32.repeat_interleave(rep, dim=1)
        v32 = v32.repeat_interleave(rep, dim=1)
    p = torch.matmul(q32, k32.transpose(2, 3)) * softmax_scale
    nq, nk = q32.shape[-2], k32.shape[-2]
    if is_causal:
        i = torch.arange(nq)[:, None]
        j = torch.arange(nk)[None, :]
        p = p.masked_fill(j > i + (nk - nq), float("-inf"))
    lse = torch.logsumexp(p, dim=-1)
    p = torch.softmax(p, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)  # empty rows
    return torch.matmul(p, v32), lse


def tiled_emulation(q, k, v, is_causal, softmax_scale, block_n=64, p_dtype=None, return_lse=False):
    """Restatement of the reference's torch tile loop flash_attention_v2
    (flash_attention_py/main_torch_only.py:160-270) generalised to (B,H,Nq,D) x (B,Hk,Nk,D):
    KV-inner loop over tiles of ``block_n`` keys (the reference's block_n = 64, :196), causal
    mask per tile (:231-237, with the C path's Nq != Nk offset attn.cpp:121-124), scale (:239),
    running max / sum (:240-257), ``local_score.to(q.dtype) @ v_tile`` (:260), final divide (:267).
    Rows are independent, so the Q-outer loop is vectorised (block_m does not change any value).
    The 16-bit rounding of P happens against the RUNNING max of each tile — the same rounding
    points as the GPU kernel, whose KV tile is also 64 keys.  Returns fp32 O (and LSE)."""
    B, H, Nq, D = q.shape
    _, Hk, Nk, _ = k.shape
    dt = q.dtype if p_dtype is None else p_dtype
    qf = q.detach().cpu().float()
    kf = k.detach().cpu().float()
    vf = v.detach().cpu().float()
    if Hk != H:
        kf = kf.repeat_interleave(H // Hk, dim=1)
        vf = vf.repeat_interleave(H // Hk, dim=1)
    g_score = torch.zeros((B, H, Nq, D), dtype=torch.float32)
    g_sum = torch.zeros((B, H, Nq, 1), dtype=torch.float32)
    g_max = torch.full((B, H, Nq, 1), -math.inf, dtype=torch.float32)
    rows = torch.arange(Nq)[:, None] + (Nk - Nq)
    for kv_start in range(0, Nk, block_n):
        k_tile = kf[:, :, kv_start:kv_start + block_n, :]
        v_tile = vf[:, :, kv_start:kv_start + block_n, :]
        qk = torch.matmul(qf, k_tile.transpose(2, 3))          # 16-bit products, fp32 accumulate
        if is_causal:
            cols = kv_start + torch.arange(k_tile.shape[2])[None, :]
            qk = qk.masked_fill(cols > rows, -math.inf)
        qk = qk * softmax_scale
        local_max = qk.max(dim=-1, keepdim=True).values
        new_max = torch.maximum(local_max, g_max)
        safe_max = torch.where(torch.isinf(new_max), torch.zeros_like(new_max), new_max)
        rescale = torch.exp(g_max - safe_max)
        local_score = torch.exp(qk - safe_max)
        g_sum = g_sum * rescale + local_score.sum(dim=-1, keepdim=True)
        g_score = g_score * rescale + torch.matmul(local_score.to(dt).float(), v_tile)
        g_max = new_max
    empty = g_sum == 0
    out = torch.where(empty, torch.zeros_like(g_score), g_score / torch.where(empty, torch.ones_like(g_sum), g_sum))
    if return_lse:
        lse = torch.where(empty, torch.full_like(g_sum, math.inf), g_max + torch.log(g_sum)).squeeze(-1)
        return out, lse
    return out


def tiled_emulation_lazy(q, k, v, is_causal, softmax_scale, block_n=64, group=32, thresh=8.0, p_dtype=None,
                         return_lse=False, key_pos=None, nk_total=None, row_pos=None):
    """The same tile loop as ``tiled_emulation`` with the issue-interleaved kernel's max rule
    (tiny-flash-attention_amd/csrc/tfa_fwd_kernel_il.h): instead of the exact running max of
    main_torch_only.py:240-257 every row keeps a REFERENCE exponent ``ref`` (log2 domain).  A group of
    ``group`` consecutive query rows (one wave) re-bases only when some row's tile max exceeds its
    ``ref`` by more than ``thresh`` (2^8): then every row of the group takes ref = max(ref, tile max) and
    O, l are multiplied by exp2(old - new).  P = exp2(s*scale*log2e - ref) <= 2^thresh, so nothing can
    overflow, and the final O = acc / l and LSE = (ref + log2 l) ln2 are unchanged mathematically; only the
    rounding points of the 16-bit P move.  With thresh=0 and group=1 this is the exact-max rule again.
    ``key_pos`` / ``nk_total``: the keys given are a SUBSET of a sequence of nk_total keys at these positions (the
    key-split kernel's wave groups); the causal mask compares positions in the whole sequence."""
    B, H, Nq, D = q.shape
    _, Hk, Nk, _ = k.shape
    dt = q.dtype if p_dtype is None else p_dtype
    qf = q.detach().cpu().float()
    kf = k.detach().cpu().float()
    vf = v.detach().cpu().float()
    if Hk != H:
        kf = kf.repeat_interleave(H // Hk, dim=1)
        vf = vf.repeat_interleave(H // Hk, dim=1)
    sc2 = torch.tensor(softmax_scale * 1.4426950408889634, dtype=torch.float32)
    ngrp = (Nq + group - 1) // group
    pad = ngrp * group - Nq
    acc = torch.zeros((B, H, Nq, D), dtype=torch.float32)
    l = torch.zeros((B, H, Nq, 1), dtype=torch.float32)
    ref = torch.full((B, H, Nq, 1), -1e30, dtype=torch.float32)
    # (row_pos: query positions of the rows when they are not 0..Nq-1 — GQA heads packed as rows, tfa_api.hip: pack_gqa_rows;
    #  nk_total - len(positions' range) is then the caller's business: pass the shift through nk_total = Nk_total_of_the_original)
    rows = (torch.arange(Nq)[:, None] + ((Nk if nk_total is None else nk_total) - Nq)) if row_pos is None else row_pos[:, None]
    pos = torch.arange(Nk) if key_pos is None else key_pos
    for kv_start in range(0, Nk, block_n):
        k_tile = kf[:, :, kv_start:kv_start + block_n, :]
        v_tile = vf[:, :, kv_start:kv_start + block_n, :]
        s = torch.matmul(qf, k_tile.transpose(2, 3))
        if is_causal:
            cols = pos[kv_start:kv_start + block_n][None, :]
            s = s.masked_fill(cols > rows, -math.inf)
        xm = s.max(dim=-1, keepdim=True).values * sc2                       # fp32 product, as the kernel
        over = xm > ref + thresh
        og = torch.nn.functional.pad(over, (0, 0, 0, pad)).view(B, H, ngrp, group).any(dim=-1, keepdim=True)
        trig = og.expand(B, H, ngrp, group).reshape(B, H, ngrp * group, 1)[:, :, :Nq]
        nref = torch.where(trig, torch.maximum(ref, xm), ref)
        alpha = torch.exp2(ref - nref)
        l = l * alpha
        acc = acc * alpha
        ref = nref
        x = (s.double() * sc2.double() - ref.double()).float()              # one rounding, like v_fma_f32
        pr = torch.exp2(x)
        l = l + pr.sum(dim=-1, keepdim=True)
        acc = acc + torch.matmul(pr.to(dt).float(), v_tile)
    empty = l == 0
    out = torch.where(empty, torch.zeros_like(acc), acc / torch.where(empty, torch.ones_like(l), l))
    if return_lse:
        lse = torch.where(empty, torch.full_like(l, math.inf), (ref + torch.log2(l)) * 0.6931471805599453).squeeze(-1)
        return out, lse
    return out


def tiled_emulation_first_tile(q, k, v, is_causal, softmax_scale, block_n=64, group=32, block_m=256, p_dtype=None,
                               return_lse=False, key_pos=None, nk_total=None, row_pos=None, return_redo=False, force_redo=None):
    """Rounding points of the MAX-FREE rule (round 6: bf16 instantiations of the il kernels that carry the hand-scheduled statement,
    tiny-flash-attention_amd/csrc/tfa_fwd_kernel_il.h MAXFREE; include/tfa.h TFA_RULE_FIRST_TILE): every row keeps the reference exponent it
    took from its FIRST key tile, ref = max_j s[i,j] * scale * log2e over that tile, for all tiles — P = exp2(s*c - ref) is rounded to bf16,
    whose exponent range is fp32's, so no running maximum is needed for range and the kernel forms none.  What it tests is what it has summed: when a
    row sum of a wave (``group`` rows) exceeds 2^40 behind a tile, the wave re-bases every row by an exact power of two (e = floor(log2 l);
    ref += e, O and l *= 2^-e — no rounding in O or l; only fl(ref + e) can move later P by ~2^-19), and when a row sum exceeds 2^64 (or is not
    finite) the workgroup REDOES that query block (``block_m`` rows): one more pass over the block's K tiles for the rows' true maxima, then the same
    pass with ref SEEDED by them (P <= 1 throughout).  The kernel's trigger looks at a lane's partial sums (between 1/8 of the row sum and all of it):
    the two can re-base at different tiles inside that band, which changes nothing but the rounding of ref + e.
    ``return_redo``: also the (B, H, blocks) mask of redone blocks; ``force_redo``: a mask to OR into it (the key-split kernel: either wave group's)."""
    B, H, Nq, D = q.shape
    _, Hk, Nk, _ = k.shape
    dt = q.dtype if p_dtype is None else p_dtype
    qf = q.detach().cpu().float()
    kf = k.detach().cpu().float()
    vf = v.detach().cpu().float()
    if Hk != H:
        kf = kf.repeat_interleave(H // Hk, dim=1)
        vf = vf.repeat_interleave(H // Hk, dim=1)
    sc2 = torch.tensor(softmax_scale * 1.4426950408889634, dtype=torch.float32)
    ngrp = (Nq + group - 1) // group
    pad = ngrp * group - Nq
    nblk = (Nq + block_m - 1) // block_m
    rows = (torch.arange(Nq)[:, None] + ((Nk if nk_total is None else nk_total) - Nq)) if row_pos is None else row_pos[:, None]
    pos = torch.arange(Nk) if key_pos is None else key_pos

    def per_group_any(x):
        og = torch.nn.functional.pad(x, (0, 0, 0, pad)).view(B, H, ngrp, group).any(dim=-1, keepdim=True)
        return og.expand(B, H, ngrp, group).reshape(B, H, ngrp * group, 1)[:, :, :Nq]

    def tile_scores(kv_start):
        s = torch.matmul(qf, kf[:, :, kv_start:kv_start + block_n, :].transpose(2, 3))
        if is_causal:
            cols = pos[kv_start:kv_start + block_n][None, :]
            s = s.masked_fill(cols > rows, -math.inf)
        return s

    def one_pass(seed):
        acc = torch.zeros((B, H, Nq, D), dtype=torch.float32)
        l = torch.zeros((B, H, Nq, 1), dtype=torch.float32)
        ref = torch.full((B, H, Nq, 1), -1e30, dtype=torch.float32)
        bad = torch.zeros((B, H, Nq, 1), dtype=torch.bool)
        for ti, kv_start in enumerate(range(0, Nk, block_n)):
            s = tile_scores(kv_start)
            if ti == 0:                                                     # the first tile's maximum (fp32 product, as the kernel); O = l = 0 so far
                ref = torch.maximum(ref, s.max(dim=-1, keepdim=True).values * sc2)
                if seed is not None:
                    ref = torch.maximum(ref, seed)
            x = (s.double() * sc2.double() - ref.double()).float()          # one rounding, like v_fma_f32
            pr = torch.exp2(x)
            l = l + pr.sum(dim=-1, keepdim=True)
            acc = acc + torch.matmul(pr.to(dt).float(), vf[:, :, kv_start:kv_start + block_n, :])
            bad = bad | ~(l <= 2.0 ** 64)
            trig = per_group_any(l > 2.0 ** 40)
            if bool(trig.any()):
                e = torch.floor(torch.log2(torch.where(torch.isfinite(l) & (l > 0), l, torch.ones_like(l)))).clamp(0, 126)
                e = torch.where(trig, e, torch.zeros_like(e))
                a = torch.exp2(-e)
                l, acc, ref = l * a, acc * a, ref + e
        empty = l == 0
        out = torch.where(empty, torch.zeros_like(acc), acc / torch.where(empty, torch.ones_like(l), l))
        lse = torch.where(empty, torch.full_like(l, math.inf), (ref + torch.log2(l)) * 0.6931471805599453).squeeze(-1)
        return out, lse, bad

    out, lse, bad = one_pass(None)
    padm = nblk * block_m - Nq
    redo = torch.nn.functional.pad(bad, (0, 0, 0, padm)).view(B, H, nblk, block_m).any(dim=-1)          # (B, H, blocks)
    if force_redo is not None:
        redo = redo | force_redo
    if bool(redo.any()):
        full = torch.full((B, H, Nq, 1), -math.inf, dtype=torch.float32)
        for kv_start in range(0, Nk, block_n):
            full = torch.maximum(full, tile_scores(kv_start).max(dim=-1, keepdim=True).values)
        o2, l2, _ = one_pass(full * sc2)
        rm = redo[..., None].expand(B, H, nblk, block_m).reshape(B, H, nblk * block_m)[:, :, :Nq]
        out = torch.where(rm[..., None], o2, out)
        lse = torch.where(rm, l2, lse)
    res = (out, lse) if return_lse else (out,)
    if return_redo:
        res = res + (redo,)
    return res if len(res) > 1 else res[0]


def ksplit_emulation(q, k, v, is_causal, softmax_scale, block_n=64, return_lse=False, rule="lazy", **kw):
    """Rounding points of the key-split kernel (VF_IL_KSPLIT in tiny-flash-attention_amd/csrc/tfa_fwd_kernel_il.h):
    two wave groups run ``tiled_emulation_lazy`` over the even and the odd ``block_n``-key tiles of the sequence (causal
    mask against the positions in the whole sequence) and the two partial results are combined by the split-KV rule
    (tiny_flash_attn.py:63-68 / README_zh.md:104-125 restated in ``merge_partials``)."""
    Nk = k.shape[2]
    nt = (Nk + block_n - 1) // block_n
    groups = []
    for g in (0, 1):
        idx = [torch.arange(t * block_n, min((t + 1) * block_n, Nk)) for t in range(g, nt, 2)]
        if idx:
            groups.append(torch.cat(idx))

    def run(emulate, **extra):
        res = [emulate(q, k.detach().cpu()[:, :, idx], v.detach().cpu()[:, :, idx], is_causal, softmax_scale, block_n,
                       return_lse=True, key_pos=idx, nk_total=Nk, **kw, **extra) for idx in groups]
        return res

    if rule == "first_tile":
        # ``rule``: each wave group keeps the maximum of ITS first tile (tile 0 / tile 1 of the head); a query block (128 rows) either group wants redone
        # is redone by both (the workgroup agrees through one LDS word), each seeded with the row maxima over its own tiles
        ft = run(tiled_emulation_first_tile, block_m=128, return_redo=True)
        redo = ft[0][2]
        for r in ft[1:]:
            redo = redo | r[2]
        if bool(redo.any()):
            ft = run(tiled_emulation_first_tile, block_m=128, return_redo=True, force_redo=redo)
        outs, lses = [r[0] for r in ft], [r[1] for r in ft]
    else:
        res = run(tiled_emulation_lazy)
        outs, lses = [r[0] for r in res], [r[1] for r in res]
    out, lse = merge_partials(torch.stack(outs), torch.stack(lses))
    return (out, lse) if return_lse else out


def partial_attn(q, k_chunk, v_chunk, is_causal, softmax_scale, kv_offset, nk_total):
    """fp32 partial attention of all queries over the key chunk [kv_offset, kv_offset+len) of a sequence of
    ``nk_total`` keys, causal mask against global positions (attn.cpp:121-124 with the chunk offset): returns
    (O_partial fp32, LSE_partial) with O = 0, LSE = +inf for rows that see no key of the chunk."""
    qf, kf, vf = (t.detach().cpu().float() for t in (q, k_chunk, v_chunk))
    H, Hk = qf.shape[1], kf.shape[1]
    if Hk != H:
        kf = kf.repeat_interleave(H // Hk, dim=1)
        vf = vf.repeat_interleave(H // Hk, dim=1)
    Nq, n = qf.shape[2], kf.shape[2]
    s = torch.matmul(qf, kf.transpose(2, 3)) * softmax_scale
    if is_causal:
        i = torch.arange(Nq)[:, None]
        j = torch.arange(n)[None, :] + kv_offset
        s = s.masked_fill(j > i + (nk_total - Nq), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0)
    o = torch.matmul(p, vf)
    lse = torch.where(torch.isneginf(lse), torch.full_like(lse, math.inf), lse)
    return o, lse


def merge_partials(o_parts, lse_parts, dtype=torch.float32):
    """The reference's v1 merge rule (flash_attention_py/tiny_flash_attn.py:63-68) in LSE form; +inf = empty part."""
    l = torch.where(torch.isposinf(lse_parts), torch.full_like(lse_parts, -math.inf), lse_parts)
    tot = torch.logsumexp(l, dim=0)
    w = torch.nan_to_num(torch.exp(l - tot), nan=0.0)
    out = (w.unsqueeze(-1) * o_parts).sum(0)
    tot = torch.where(torch.isneginf(tot), torch.full_like(tot, math.inf), tot)
    return out.to(dtype), tot


def attn_bwd_reference(q, k, v, dout, is_causal, softmax_scale, dtype=torch.float64):
    """Gradients of O = softmax(scale * Q K^T + mask) V w.r.t. q, k, v by autograd on the CPU in ``dtype``
    (fp64 = ground truth) — the function the reference's forward implements (attn.cpp:35-98) differentiated;
    the reference itself has no backward (it only saves the LSE for one, flash_attention.cu:353-354).
    q (B,H,Nq,D), k/v (B,Hk,Nk,D) with H % Hk == 0 (dk, dv are summed over the query heads of a kv head);
    causal mask bottom-right aligned (attn.cpp:121-124).  Returns (dq, dk, dv) in ``dtype``."""
    qf = q.detach().cpu().to(dtype).requires_grad_(True)
    kf = k.detach().cpu().to(dtype).requires_grad_(True)
    vf = v.detach().cpu().to(dtype).requires_grad_(True)
    B, H, Nq, D = qf.shape
    Hk, Nk = kf.shape[1], kf.shape[2]
    ke = kf.repeat_interleave(H // Hk, dim=1) if Hk != H else kf
    ve = vf.repeat_interleave(H // Hk, dim=1) if Hk != H else vf
    s = torch.matmul(qf, ke.transpose(2, 3)) * softmax_scale
    if is_causal:
        i = torch.arange(Nq)[:, None]
        j = torch.arange(Nk)[None, :]
        s = s.masked_fill(j > i + (Nk - Nq), float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)      # rows with no visible key
    o = torch.matmul(p, ve)
    o.backward(dout.detach().cpu().to(dtype))
    return qf.grad, kf.grad, vf.grad


def attn_bwd_bounds(q, k, v, out, dout, is_causal, softmax_scale):
    """Non-cancelling magnitudes of the three gradients, for error bounds like ``abs_weighted`` gives for O:
    rounding P (for dV) and dS (for dQ, dK) to 16 bit perturbs each term of the sums by a relative 2^-9 (bf16)
    / 2^-12 (fp16), and delta_i = sum_d dO_id O_id is formed from the 16-bit O the forward stored, which
    moves every dS_ij by P_ij * 2^-9 * sum_d |dO_id O_id| (inherent to the algorithm: it re-uses the saved O).
    Returns (Aq, Ak, Av) in fp64, shaped like q, k, v:  |grad_kernel - grad_exact| <= eps16 * A (+ final rounding)."""
    qf, kf, vf = (t.detach().cpu().double() for t in (q, k, v))
    of, dof = out.detach().cpu().double(), dout.detach().cpu().double()
    B, H, Nq, D = qf.shape
    Hk, Nk = kf.shape[1], kf.shape[2]
    G = H // Hk
    ke = kf.repeat_interleave(G, dim=1) if G > 1 else kf
    ve = vf.repeat_interleave(G, dim=1) if G > 1 else vf
    s = torch.matmul(qf, ke.transpose(2, 3)) * softmax_scale
    if is_causal:
        i = torch.arange(Nq)[:, None]
        j = torch.arange(Nk)[None, :]
        s = s.masked_fill(j > i + (Nk - Nq), float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0)
    dp = torch.matmul(dof, ve.transpose(2, 3))
    delta = (dof * of).sum(-1, keepdim=True)
    ds_abs = p * ((dp - delta).abs() + (dof * of).abs().sum(-1, keepdim=True))   # |dS| + its sensitivity to the rounded O
    aq = softmax_scale * torch.matmul(ds_abs, ke.abs())
    ak = softmax_scale * torch.matmul(ds_abs.transpose(2, 3), qf.abs())
    av = torch.matmul(p.transpose(2, 3), dof.abs())
    if G > 1:
        ak = ak.view(B, Hk, G, Nk, D).sum(2)
        av = av.view(B, Hk, G, Nk, D).sum(2)
    return aq, ak, av


def abs_weighted(q, k, v, is_causal, softmax_scale):
    """A[i,d] = sum_j P[i,j] |v[j,d]| — the non-cancelling magnitude of each output element; the
    natural scale for error bounds on O (|O| <= A, with equality when no cancellation)."""
    return exact64(q, k, v.abs(), is_causal, softmax_scale)


def tiny_py_multihead(q, k, v, block_m=4):
    """Restatement of the reference's pure-Python path, BASELINE config 1
    (flash_attention_py/tiny_flash_attn.py:137-196 flash_attn_v2_multihead): a Python double loop over
    block_m-row query blocks and block_m-key blocks of fp32 (B,H,N,D) CPU tensors, online softmax with a
    running max / denominator per row, NO softmax scale and NO mask (:168-170), floor(N / block_m) blocks
    (:154,164; trailing rows stay zero).  bench.py times this as the "Python CPU path" next to the GPU number;
    tests/test_oracle.py pins it to the reference's own output (tests/golden/tiny_py_cfg1.npz)."""
    q, k, v = q.float(), k.float(), v.float()
    B, H, N, _ = q.shape
    out = torch.zeros_like(v)
    nblk = N // block_m
    for jb in range(nblk):
        rows = slice(jb * block_m, (jb + 1) * block_m)
        qb = q[..., rows, :]
        acc = out[..., rows, :]
        den = torch.zeros((B, H, block_m, 1))
        mx = torch.full((B, H, block_m, 1), -math.inf)
        for ib in range(k.shape[-2] // block_m):
            keys = slice(ib * block_m, (ib + 1) * block_m)
            s = qb @ k[..., keys, :].transpose(2, 3)
            mx_new = torch.maximum(mx, s.max(dim=-1, keepdim=True).values)
            e = torch.exp(s - mx_new)
            alpha = torch.exp(mx - mx_new)
            den = den * alpha + e.sum(dim=-1, keepdim=True)
            acc = acc * alpha + e @ v[..., keys, :]
            mx = mx_new
        out[..., rows, :] = acc / den
    return out


_ref_mod = None


def ref_kernels():
    """The reference's own compiled CPU module ``_kernels`` (naive_attn / flash_attn), or None
    when oracle/_ref has not been built (it can only be built where /root/reference exists)."""
    global _ref_mod
    if _ref_mod is None:
        cands = glob.glob(os.path.join(_HERE, "_ref", "_kernels*.so"))
        if not cands:
            return None
        # a qualified name: the GPU product ships its own module called `_kernels` (the reference's name), and CPython hands
        # back an already-loaded extension module of the same name instead of loading this file
        spec = importlib.util.spec_from_file_location("tfa_oracle_ref._kernels", cands[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _ref_mod = mod
    return _ref_mod


def make_inputs(B, H, N, D, dtype=torch.float16, seed=0, std=0.5, Hk=None, Nk=None, dist="normal"):
    """The reference's input recipe (flash_attention_cutlass/test.py:13-17: normal(0, 0.5) cast to
    the dtype; flash_attention_c/test.py:35-40: uniform [0,1) fp32), generated on CPU from a seed
    so the oracle and the GPU see identical bits."""
    g = torch.Generator().manual_seed(seed)
    Hk = H if Hk is None else Hk
    Nk = N if Nk is None else Nk

    def mk(shape):
        if dist == "normal":
            return torch.empty(shape, dtype=torch.float32).normal_(0.0, std, generator=g).to(dtype)
        return torch.rand(shape, dtype=torch.float32, generator=g).to(dtype)

    return mk((B, H, N, D)), mk((B, Hk, Nk, D)), mk((B, Hk, Nk, D))
