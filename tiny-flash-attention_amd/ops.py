"""Reference-named operators over the C ABI (include/tfa.h).

Argument meaning, order, return shapes and error behaviour follow the reference's pybind
functions; the compute is the hand-written HIP kernel in csrc/.  Nothing here falls back to
PyTorch or the CPU: if the library is missing or the device is not a GPU the call raises.
"""
import ctypes as C
import math

import torch

from . import _lib

_DT = {torch.float16: _lib.TFA_F16, torch.bfloat16: _lib.TFA_BF16}
_DT_IN = dict(_DT)
_DT_IN[torch.float32] = _lib.TFA_F32    # fp32 q,k,v: the fp32 correctness path (the reference's fp32 fixtures; tfa_fwd_f32.hip) — forward only


def _dt16(t, what):
    """dtype code of a float16 / bfloat16 tensor for the entries that have no fp32 path (backward, split-KV, raw parameter blocks)."""
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"{what}: float16 or bfloat16 only (got {t.dtype}); the fp32 path is forward-only — ops.flash_attn_fwd / tfa_fwd "
                        "with dtype = TFA_F32") from None


def _check_input(x, name):
    # CHECK_INPUT of the reference (flash_attention_cutlass/include/attention_api.cuh:12-18)
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _strides_bhnd(t, layout):
    """(batch, head, row) element strides of a 4-D tensor given its logical layout."""
    if layout == "bhnd":
        return t.stride(0), t.stride(1), t.stride(2)
    if layout == "bnhd":
        return t.stride(0), t.stride(2), t.stride(1)
    raise ValueError(f"unknown layout {layout!r}")


def flash_attn_fwd(q, k, v, is_causal=False, softmax_scale=None, *, layout="bhnd", out_f32=False,
                   return_lse=True, out=None, kv_offset=0, nk_total=None, auto_split=False, exact_max=False):
    """General forward: q (B,H,Nq,D) / k,v (B,Hk,Nk,D) for ``layout='bhnd'`` or
    (B,N,H,D) for ``layout='bnhd'``; any batch/head/row strides, unit stride along D.
    Returns ``(out, lse)``; ``out`` has q's shape (fp32 when ``out_f32``), ``lse`` is (B,H,Nq) fp32.
    Maps onto tfa_fwd (include/tfa.h); semantics per flash_attention_c/csrc/attn.cpp:101-169 and
    flash_attention_cutlass/csrc/flash_attention.cu:536-630.  ``auto_split``: decode-like shapes (few query rows, long K/V)
    go through tfa_fwd_splitkv with the chunk count tfa_fwd_suggest_splits names (what the reference-named entry points do).
    ``exact_max``: TFA_FWD_EXACT_MAX — P is rounded to 16 bits at the reference's own points (exact running row maximum per
    KV tile, main_torch_only.py:240-260); head dims up to 128."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
        if t.dim() != 4:
            raise RuntimeError(f"{n} must be 4-D")
        if t.stride(3) != 1:
            raise RuntimeError(f"{n} must have unit stride along the head dimension")
    if q.dtype not in _DT_IN or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError(f"q,k,v must share dtype float16, bfloat16 or float32 (got {q.dtype}, {k.dtype}, {v.dtype})")
    if q.dtype == torch.float32 and (exact_max or (out is not None and out.dtype != torch.float32)):
        raise TypeError("float32 q,k,v: the fp32 correctness path returns float32 and has no 16-bit rounding points (exact_max does not apply)")
    if k.device != q.device or v.device != q.device:
        raise RuntimeError("q,k,v must be on the same device")
    if layout == "bhnd":
        B, H, Nq, D = q.shape
        Bk, Hk, Nk, Dk = k.shape
    else:
        B, Nq, H, D = q.shape
        Bk, Nk, Hk, Dk = k.shape
    if k.shape != v.shape or Bk != B or Dk != D:
        raise RuntimeError(f"shape mismatch: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}")
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(D)

    if out is None:
        out = torch.empty(q.shape, dtype=torch.float32 if out_f32 else q.dtype, device=q.device)
    else:   # a caller-owned result buffer: the kernel writes q.shape elements at out's strides — check before launching
        if not isinstance(out, torch.Tensor) or out.shape != q.shape:
            raise RuntimeError(f"out must be a tensor shaped like q {tuple(q.shape)}")
        if out.device != q.device:
            raise RuntimeError("out must be on q's device")
        if out.dtype not in (q.dtype, torch.float32):
            raise RuntimeError(f"out must be {q.dtype} or float32 (got {out.dtype})")
        if out.stride(3) != 1:
            raise RuntimeError("out must have unit stride along the head dimension")
    lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device) if return_lse else None

    p = _lib.TfaFwdParams()
    p.q, p.k, p.v, p.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.lse = lse.data_ptr() if lse is not None else None
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", out)):
        s = _strides_bhnd(t, layout)
        arr = getattr(p, name)
        arr[0], arr[1], arr[2] = s
    p.softmax_scale = float(softmax_scale)
    p.is_causal = 1 if is_causal else 0
    p.dtype = _DT_IN[q.dtype]
    p.out_dtype = _lib.TFA_F32 if out.dtype == torch.float32 else _DT[out.dtype]
    p.kv_offset = int(kv_offset)                     # split-KV: k, v are keys [kv_offset, kv_offset+Nk) of nk_total
    p.nk_total = 0 if nk_total is None else int(nk_total)
    p.flags = _lib.TFA_FWD_EXACT_MAX if exact_max else 0
    L = _lib.lib()
    # tfa_fwd_splitkv's merge writes a dense (B,H,Nq,D) result: gate on the exact strides it checks (is_contiguous() ignores the
    # strides of size-1 dims, and Nq == 1 is the very shape auto-split targets)
    dense_out = (out.stride(3) == 1 and out.stride(2) == D and out.stride(1) == Nq * D and out.stride(0) == H * Nq * D)
    splits = L.tfa_fwd_suggest_splits(C.byref(p)) if (auto_split and layout == "bhnd" and dense_out) else 1
    with torch.cuda.device(q.device):
        stream = torch.cuda.current_stream().cuda_stream
        if splits > 1:
            need = L.tfa_fwd_splitkv_workspace(C.byref(p), int(splits))
            if need < 0:
                _lib.check(int(need))
            ws = torch.empty((int(need),), dtype=torch.float32, device=q.device)
            _lib.check(L.tfa_fwd_splitkv(C.byref(p), int(splits), ws.data_ptr(), C.c_void_p(stream)))
        else:
            _lib.check(L.tfa_fwd(C.byref(p), C.c_void_p(stream)))
    return out, lse


def merge_partials(o_parts, lse_parts, out_dtype=torch.bfloat16):
    """Split-KV merge (tfa_merge): ``o_parts`` (P,B,H,Nq,D) fp32 and ``lse_parts`` (P,B,H,Nq) fp32 are partial
    attention results of the same queries over disjoint key chunks (``flash_attn_fwd(..., out_f32=True,
    kv_offset=..., nk_total=...)``); returns ``(out, lse)`` of the whole key sequence.  The rule is the reference's
    v1 merge (flash_attention_py/tiny_flash_attn.py:63-68) in LSE form."""
    if not o_parts.is_cuda or o_parts.dtype != torch.float32 or lse_parts.dtype != torch.float32:
        raise RuntimeError("o_parts / lse_parts must be fp32 CUDA tensors")
    o_parts, lse_parts = o_parts.contiguous(), lse_parts.contiguous()
    P, D = o_parts.shape[0], o_parts.shape[-1]
    rows = lse_parts[0].numel()
    if o_parts[0].numel() != rows * D or lse_parts.shape[0] != P:
        raise RuntimeError("shape mismatch between o_parts and lse_parts")
    out = torch.empty(o_parts.shape[1:], dtype=out_dtype, device=o_parts.device)
    lse = torch.empty(lse_parts.shape[1:], dtype=torch.float32, device=o_parts.device)
    code = _lib.TFA_F32 if out_dtype == torch.float32 else _DT[out_dtype]
    with torch.cuda.device(o_parts.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().tfa_merge(o_parts.data_ptr(), lse_parts.data_ptr(), P, rows, D, rows * D, rows,
                                        out.data_ptr(), code, lse.data_ptr(), C.c_void_p(stream)))
    return out, lse


def flash_attn_fwd_splitkv(q, k, v, is_causal=False, softmax_scale=None, *, splits=2, return_lse=True, native=True):
    """The forward as `splits` partial passes over contiguous key chunks (multiples of 64 keys) + tfa_merge;
    (B,H,N,D) layout, out contiguous.  ``native=True``: tfa_fwd_splitkv — ONE launch whose grid carries a copy of the
    work per chunk (what decode-like shapes need to fill the chip), partials in a scratch tensor.
    ``native=False``: one tfa_fwd call per chunk driven from Python (what dist.kv_sharded_forward does per rank)."""
    Nk = k.shape[2]
    if native:
        B, H, Nq, D = q.shape
        if softmax_scale is None:
            softmax_scale = 1.0 / math.sqrt(D)
        out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        p = make_params(q, k, v, out, lse, is_causal, softmax_scale)
        L = _lib.lib()
        need = L.tfa_fwd_splitkv_workspace(C.byref(p), int(splits))
        if need < 0:
            _lib.check(int(need))
        ws = torch.empty((int(need),), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            stream = torch.cuda.current_stream().cuda_stream
            _lib.check(L.tfa_fwd_splitkv(C.byref(p), int(splits), ws.data_ptr(), C.c_void_p(stream)))
        return (out, lse) if return_lse else out
    step = ((Nk + splits - 1) // splits + 63) // 64 * 64
    o_parts, l_parts = [], []
    for lo in range(0, Nk, step):
        hi = min(Nk, lo + step)
        o, l = flash_attn_fwd(q, k[:, :, lo:hi], v[:, :, lo:hi], is_causal, softmax_scale, out_f32=True, kv_offset=lo, nk_total=Nk)
        o_parts.append(o)
        l_parts.append(l)
    out, lse = merge_partials(torch.stack(o_parts), torch.stack(l_parts), q.dtype)
    return (out, lse) if return_lse else out


def make_params(q, k, v, out, lse, is_causal, softmax_scale, layout="bhnd"):
    """Build a TfaFwdParams for existing buffers (used by bench.py / tfa_fwd_time)."""
    p = _lib.TfaFwdParams()
    p.q, p.k, p.v, p.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.lse = lse.data_ptr() if lse is not None else None
    if layout == "bhnd":
        B, H, Nq, D = q.shape
        _, Hk, Nk, _ = k.shape
    else:
        B, Nq, H, D = q.shape
        _, Nk, Hk, _ = k.shape
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", out)):
        s = _strides_bhnd(t, layout)
        arr = getattr(p, name)
        arr[0], arr[1], arr[2] = s
    p.softmax_scale = float(softmax_scale)
    p.is_causal = 1 if is_causal else 0
    p.dtype = _dt16(q, "make_params (q, k, v)")
    p.out_dtype = _lib.TFA_F32 if out.dtype == torch.float32 else _dt16(out, "make_params (out)")
    return p


def make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, is_causal, softmax_scale, layout="bhnd"):
    """Build a TfaBwdParams for existing buffers."""
    p = _lib.TfaBwdParams()
    p.q, p.k, p.v, p.out, p.dout = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr()
    p.lse, p.delta = lse.data_ptr(), delta.data_ptr()
    p.dq, p.dk, p.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    if layout == "bhnd":
        B, H, Nq, D = q.shape
        _, Hk, Nk, _ = k.shape
    else:
        B, Nq, H, D = q.shape
        _, Nk, Hk, _ = k.shape
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", out), ("do_stride", dout),
                    ("dq_stride", dq), ("dk_stride", dk), ("dv_stride", dv)):
        s = _strides_bhnd(t, layout)
        arr = getattr(p, name)
        arr[0], arr[1], arr[2] = s
    p.softmax_scale = float(softmax_scale)
    p.is_causal = 1 if is_causal else 0
    p.dtype = _dt16(q, "backward (q, k, v)")
    p.grad_dtype = _lib.TFA_F32 if dq.dtype == torch.float32 else _dt16(dq, "backward (gradients)")
    return p


def flash_attn_bwd(q, k, v, out, lse, dout, is_causal=False, softmax_scale=None, *, layout="bhnd", grad_f32=False, workspace=None):
    """Backward of ``flash_attn_fwd``: returns ``(dq, dk, dv)`` shaped like q, k, v (fp32 when ``grad_f32``).
    ``out`` and ``lse`` are the forward's results for the same q, k, v; ``dout`` is the upstream gradient
    (shape/dtype of ``out``).  The reference has no backward — it only saves the LSE for one
    (flash_attention_cutlass/csrc/flash_attention.cu:353-354,614-623); maps onto tfa_bwd (include/tfa.h).
    ``workspace``: None (default: the O(N)-memory 7-GEMM form), True (allocate tfa_bwd_workspace_bytes of scratch for this call)
    or a caller-owned uint8 / any-dtype CUDA tensor of at least that many bytes: tfa_bwd then keeps dS and executes 5 GEMMs."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout")):
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
        if t.dim() != 4 or t.stride(3) != 1:
            raise RuntimeError(f"{n} must be 4-D with unit stride along the head dimension")
    if q.dtype not in _DT or any(t.dtype != q.dtype for t in (k, v, out, dout)):
        raise TypeError("q,k,v,out,dout must share dtype float16 or bfloat16")
    if out.shape != q.shape or dout.shape != q.shape or k.shape != v.shape:
        raise RuntimeError("shape mismatch")
    D = q.shape[-1]
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(D)
    lse_shape = (q.shape[0], q.shape[1], q.shape[2]) if layout == "bhnd" else (q.shape[0], q.shape[2], q.shape[1])   # (B,H,Nq)
    if not lse.is_cuda or lse.device != q.device or lse.dtype != torch.float32 or tuple(lse.shape) != lse_shape:
        raise RuntimeError(f"lse must be the forward's float32 {lse_shape} tensor on q's device "
                           f"(got {lse.dtype} {tuple(lse.shape)} on {lse.device})")
    if any(t.device != q.device for t in (k, v, out, dout)):
        raise RuntimeError("q,k,v,out,dout must be on the same device")
    lse = lse.contiguous()
    gdt = torch.float32 if grad_f32 else q.dtype
    dq = torch.empty(q.shape, dtype=gdt, device=q.device)
    dk = torch.empty(k.shape, dtype=gdt, device=q.device)
    dv = torch.empty(v.shape, dtype=gdt, device=q.device)
    delta = torch.empty(lse.shape, dtype=torch.float32, device=q.device)
    p = make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, is_causal, softmax_scale, layout)
    if workspace is not None and workspace is not False:
        need = bwd_workspace_bytes(p)
        if workspace is True:
            workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=q.device)
        if not isinstance(workspace, torch.Tensor) or not workspace.is_cuda or workspace.device != q.device or not workspace.is_contiguous():
            raise RuntimeError("workspace must be True or a contiguous CUDA tensor on q's device")
        p.workspace = workspace.data_ptr()
        p.workspace_bytes = workspace.numel() * workspace.element_size()
    with torch.cuda.device(q.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().tfa_bwd(C.byref(p), C.c_void_p(stream)))
    return dq, dk, dv


def bwd_workspace_bytes(p):
    """tfa_bwd_workspace_bytes for a TfaBwdParams (0: the 5-GEMM form does not apply to this problem)."""
    L = _lib.lib()
    L.tfa_bwd_workspace_bytes.restype = C.c_longlong
    n = L.tfa_bwd_workspace_bytes(C.byref(p))
    if n < 0:
        _lib.check(int(n))
    return int(n)


# ---------------------------------------------------------------------------------------------
# the reference's three operator entry points
# ---------------------------------------------------------------------------------------------
def flash_attention_v2_cutlass(q, k, v, is_causal, softmax_scale):
    """``attention_cutlass.flash_attention_v2_cutlass(q, k, v, is_causal, softmax_scale) -> [out, lse]``
    (flash_attention_cutlass/csrc/flash_attention.cu:741-772).  q,k,v: contiguous CUDA
    (B,H,N,D) fp16/bf16; out like q; lse (B,H,N) fp32.  All five arguments are positional, as
    in the reference binding (no py::arg, attention_api.cpp:6-10).  Launches on the current
    stream and does not synchronise (the reference blocks; its callers synchronise anyway)."""
    _check_input(q, "q")
    _check_input(k, "k")
    _check_input(v, "v")
    out, lse = flash_attn_fwd(q, k, v, bool(is_causal), float(softmax_scale), auto_split=True)
    return [out, lse]


def flash_attention_v2_cuda(q, k, v):
    """``attention_cuda.flash_attention_v2_cuda(q, k, v) -> out``: non-causal, scale fixed to
    1/sqrt(D) inside (flash_attention_cuda/csrc/flash_attention.cu:375-424, :389)."""
    _check_input(q, "q")
    _check_input(k, "k")
    _check_input(v, "v")
    out, _ = flash_attn_fwd(q, k, v, False, 1.0 / math.sqrt(q.shape[-1]), return_lse=False, auto_split=True)
    return out


# the reference exports three names from attention_cuda; all compute the same function
# (flash_attention_cuda/csrc/attention_api.cpp:6-14) — here they share the one kernel.
flash_attention_v1_cuda = flash_attention_v2_cuda
self_attention_cuda = flash_attention_v2_cuda


def flash_attn(q, k, v, is_causal, softmax_scale):
    """``_kernels.flash_attn(q, k, v, is_causal, softmax_scale) -> out``
    (flash_attention_c/csrc/attn.cpp:237-262).  Same math as the CPU sibling, including its
    bottom-right-aligned causal mask for Nq != Nk (attn.cpp:121-124) and strided inputs
    (attn.cpp:171-203); tensors live on the GPU: fp16 / bf16 (the MFMA kernels) or fp32 (the reference's own fixture dtype: the fp32
    correctness path, fp32 arithmetic end to end)."""
    out, _ = flash_attn_fwd(q, k, v, bool(is_causal), float(softmax_scale), return_lse=False, auto_split=True)
    return out


# naive_attn computes the same function by a different route in the reference
# (attn.cpp:35-98); the GPU path has one kernel.
naive_attn = flash_attn


class _FlashAttnBNHD(torch.autograd.Function):
    """autograd glue for ``flash_attn_func``: forward = tfa_fwd, backward = tfa_bwd, both on (B,N,H,D) views."""

    @staticmethod
    def forward(ctx, q, k, v, causal, softmax_scale):
        out, lse = flash_attn_fwd(q, k, v, causal, softmax_scale, layout="bnhd")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.causal, ctx.scale = causal, softmax_scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        if dout.stride(3) != 1:
            dout = dout.contiguous()
        dq, dk, dv = flash_attn_bwd(q, k, v, out, lse, dout, ctx.causal, ctx.scale, layout="bnhd")
        return dq, dk, dv, None, None


def flash_attn_func(q, k, v, causal=False, softmax_scale=None):
    """(B,N,H,D)-layout entry with the signature the reference's scripts use for comparison
    (flash_attention_cutlass/test.py:71-76, flash_attention_py/main_torch_only.py:304);
    supports GQA/MQA (fewer K/V heads).  Differentiable (like the official function the reference
    compares against): when an input requires grad the backward runs tfa_bwd."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _FlashAttnBNHD.apply(q, k, v, bool(causal), softmax_scale)
    out, _ = flash_attn_fwd(q, k, v, causal, softmax_scale, layout="bnhd", return_lse=False)
    return out
