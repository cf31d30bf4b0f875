"""GPU parity tests of the backward pass (tfa_bwd through the C ABI) against the fp64 autograd oracle.

The reference has no backward (it only saves the LSE for one, flash_attention.cu:353-354,614-623), so the
oracle is the function the reference's forward implements (attn.cpp:35-98), differentiated by autograd in fp64
on identical seeded inputs (oracle.attn_bwd_reference).  Tolerances (floating point):
  (B1) fp32-gradient debug path: |d| <= eps16 * A + 1e-6 with eps16 = 2^-8 (bf16) / 2^-11 (fp16) and
       A = oracle.attn_bwd_bounds: the non-cancelling magnitude of each gradient element, i.e. the rigorous
       bound for rounding P and dS to 16 bit (what FlashAttention-2 does) and for forming delta from the
       16-bit O the forward stored.
  (B2) 16-bit gradients: (B1) plus half an ulp of the result.
  (B3) against the reference's own bar for its forward (atol 1e-2, test.py:87), scaled with the gradient
       magnitude: max|d| <= 1e-2 * max(1, max|ref|).
"""
import math

import pytest
import torch

from helpers import ulp16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tfa():
    import tiny_flash_attention_amd as m
    from tiny_flash_attention_amd import _lib

    _lib.lib()
    return m


def make_dout(B, H, Nq, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.empty((B, H, Nq, D), dtype=torch.float32).normal_(0.0, 0.5, generator=g).to(dtype)


def check_grads(oracle, grads32, grads16, q, k, v, out16, dout, causal, sc, dtype):
    ref = oracle.attn_bwd_reference(q, k, v, dout, causal, sc)
    bounds = oracle.attn_bwd_bounds(q, k, v, out16, dout, causal, sc)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for name, g32, g16, r, A in zip(("dq", "dk", "dv"), grads32, grads16, ref, bounds):
        g32c, g16c = g32.double().cpu(), g16.double().cpu()
        assert bool(torch.isfinite(g16c).all()), f"{name}: non-finite"
        ex = ((g32c - r).abs() - (eps * A + 1e-6)).max().item()
        assert ex <= 0, f"(B1) {name}: fp32 gradient exceeds the 16-bit rounding bound by {ex:.3e}"
        ex16 = ((g16c - r).abs() - (eps * A + 0.5 * ulp16(r.float(), dtype).double() * (1 + 1e-3) + 1e-6)).max().item()
        assert ex16 <= 0, f"(B2) {name}: 16-bit gradient exceeds bound + half an ulp by {ex16:.3e}"
        assert (g16c - r).abs().max().item() <= 1e-2 * max(1.0, r.abs().max().item()), f"(B3) {name}"


MODES = ["default", "workspace", "split"]   # tfa_bwd's forms: 7 GEMM units (fused dK/dV launch), 5 (dS kept in a caller-lent
                                            # workspace, dQ = dS.K), 8 (dK and dV as two launches: the debug A/B arm)


def run_bwd_case(tfa, oracle, dev, dtype, B, H, Hk, Nq, Nk, D, causal, seed=0, scale=None, layout="bhnd", mode="default"):
    from tiny_flash_attention_amd import _lib, ops

    q, k, v = oracle.make_inputs(B, H, Nq, D, dtype, seed=seed, Hk=Hk, Nk=Nk)
    dout = make_dout(B, H, Nq, D, dtype, seed + 100)
    sc = 1.0 / math.sqrt(D) if scale is None else scale
    qd, kd, vd, dod = q.to(dev), k.to(dev), v.to(dev), dout.to(dev)
    if layout == "bnhd":
        qd, kd, vd, dod = (t.transpose(1, 2).contiguous() for t in (qd, kd, vd, dod))
    out, lse = ops.flash_attn_fwd(qd, kd, vd, causal, sc, layout=layout)
    ws = True if mode == "workspace" else None
    _lib.debug_bwd_split(mode == "split")
    try:
        g32 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, sc, layout=layout, grad_f32=True, workspace=ws)
        g16 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, sc, layout=layout, workspace=ws)
        torch.cuda.synchronize()
    finally:
        _lib.debug_bwd_split(False)
    if layout == "bnhd":
        g32 = tuple(t.transpose(1, 2) for t in g32)
        g16 = tuple(t.transpose(1, 2) for t in g16)
        out = out.transpose(1, 2)
    check_grads(oracle, g32, g16, q, k, v, out.cpu(), dout, causal, sc, dtype)
    return g16


BWD_SHAPES = [
    # dtype, B, H, Hk, Nq, Nk, D, causal
    (torch.bfloat16, 1, 2, 2, 256, 256, 128, False),
    (torch.bfloat16, 1, 2, 2, 256, 256, 128, True),
    (torch.float16, 2, 4, 2, 320, 320, 64, True),        # GQA, N not a multiple of the resident block
    (torch.bfloat16, 1, 4, 1, 200, 333, 128, True),      # MQA, ragged, Nq < Nk (bottom-right causal)
    (torch.float16, 1, 2, 2, 333, 200, 64, False),       # Nq > Nk
    (torch.float16, 1, 3, 3, 700, 100, 128, True),       # causal with 600 query rows that see no key
    (torch.bfloat16, 2, 2, 2, 1024, 1024, 128, True),
    # the hand-scheduled statements of both launches (128 wide; tools/gen_bwd_dq_asm_loop.py, gen_bwd_kv_asm_loop.py): whole tiles, several per block, the diagonal's masked
    # bodies, per streamed head under GQA / MQA, both 16-bit types, bottom-right causal with Nq != Nk
    (torch.bfloat16, 1, 4, 2, 1024, 1024, 128, True),
    (torch.float16, 1, 4, 1, 768, 1280, 128, False),
    (torch.float16, 1, 2, 2, 1024, 896, 128, True),
    (torch.bfloat16, 1, 6, 2, 640, 1152, 128, True),
    (torch.bfloat16, 1, 1, 1, 64, 64, 128, False),       # a single tile
    (torch.float16, 1, 2, 1, 1, 500, 64, True),          # single query row (decode-like)
    (torch.bfloat16, 1, 1, 1, 77, 65, 64, False),
    (torch.bfloat16, 3, 2, 2, 513, 513, 64, True),
    (torch.bfloat16, 1, 2, 2, 300, 300, 96, True),       # head dims inside the 128- and 64-wide kernels (columns beyond D read as zeros)
    (torch.float16, 2, 2, 1, 200, 333, 32, False),
    (torch.bfloat16, 1, 1, 1, 130, 130, 72, True),
    # head dims above 128 (the reference's buckets 160 / 192 / 224 / 256, static_switch.h:39-66): the 256-wide single-gradient launches
    (torch.bfloat16, 1, 2, 2, 384, 384, 256, True),
    (torch.float16, 2, 4, 2, 200, 333, 160, True),       # GQA, ragged, bottom-right causal
    (torch.bfloat16, 1, 2, 1, 300, 130, 192, False),     # MQA, Nq > Nk
    (torch.float16, 1, 1, 1, 513, 513, 224, True),
]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dtype,B,H,Hk,Nq,Nk,D,causal", BWD_SHAPES)
def test_bwd_parity(tfa, oracle, dev, dtype, B, H, Hk, Nq, Nk, D, causal, mode):
    run_bwd_case(tfa, oracle, dev, dtype, B, H, Hk, Nq, Nk, D, causal, seed=3, mode=mode)


@pytest.mark.parametrize("mode", ["default", "workspace"])
def test_bwd_strided_bnhd_matches_bhnd(tfa, oracle, dev, mode):
    a = run_bwd_case(tfa, oracle, dev, torch.bfloat16, 2, 8, 2, 384, 384, 128, True, seed=5, scale=0.09, mode=mode)
    b = run_bwd_case(tfa, oracle, dev, torch.bfloat16, 2, 8, 2, 384, 384, 128, True, seed=5, scale=0.09, layout="bnhd", mode=mode)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_bwd_workspace_form_agrees_with_the_default(tfa, oracle, dev):
    """dK and dV come from the same launch in both forms — on the same delta up to fp32 summation order (the default form computes delta inside
    its dQ launch, the workspace form by a launch of its own: a last-bit difference of delta may move a 16-bit gradient by one ulp); dQ from
    the kept dS differs from the recomputing launch only by P having been rounded to 16 bit before dS was formed: inside the (B1) bound,
    and deterministic."""
    from tiny_flash_attention_amd import ops

    q, k, v = (t.to(dev) for t in oracle.make_inputs(2, 8, 1280, 128, torch.bfloat16, seed=21, Hk=4, Nk=1536))
    dout = make_dout(2, 8, 1280, 128, torch.bfloat16, 22).to(dev)
    out, lse = ops.flash_attn_fwd(q, k, v, True, 0.09)
    g0 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, 0.09)
    big = torch.full((64 << 20,), 0xFF, dtype=torch.uint8, device=dev)          # a caller-owned scratch buffer full of NaN patterns
    g1 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, 0.09, workspace=big)
    g2 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, 0.09, workspace=big)
    torch.cuda.synchronize()
    assert torch.equal(g0[2], g1[2])                  # dV does not depend on delta
    dkd = (g0[1].float() - g1[1].float()).abs()
    assert bool((dkd <= 2.0 * ulp16(g0[1].float(), torch.bfloat16) + 1e-3 * g0[1].float().abs().max()).all()), dkd.max().item()
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    d = (g0[0].float() - g1[0].float()).abs().max().item()
    assert bool(torch.isfinite(g1[0].float()).all()) and d <= 2e-2 * g0[0].float().abs().max().item(), d
    small = torch.empty((1024,), dtype=torch.uint8, device=dev)                  # too small: silently the O(N)-memory form
    g3 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, 0.09, workspace=small)
    torch.cuda.synchronize()
    for a, b in zip(g0, g3):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,B,H,Hk,Nq,Nk,D,causal", [(torch.bfloat16, 2, 8, 4, 1000, 1280, 128, True), (torch.float16, 1, 4, 4, 777, 777, 64, False),
                                                        (torch.bfloat16, 1, 2, 1, 300, 520, 256, True), (torch.float16, 1, 6, 2, 513, 513, 96, True)])
def test_bwd_delta_inside_the_dq_launch(tfa, oracle, dev, dtype, B, H, Hk, Nq, Nk, D, causal):
    """tfa_bwd computes delta = rowsum(dO o O) inside its dQ launch (round 4: one launch and one pass over dO less); tfa_debug_bwd_split
    bit 3 restores the launch of its own.  Same delta up to fp32 summation order, written to tfa_bwd_params::delta in both forms (every
    row, ragged last block included); gradients within an ulp of each other."""
    import ctypes as C
    from tiny_flash_attention_amd import _lib, ops

    q, k, v = (t.to(dev) for t in oracle.make_inputs(B, H, Nq, D, dtype, seed=33, Hk=Hk, Nk=Nk))
    dout = make_dout(B, H, Nq, D, dtype, 34).to(dev)
    sc = 1.0 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(q, k, v, causal, sc)
    res = []
    for flag in (0, 8):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.full_like(lse, float("nan"))
        pb = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
        _lib.debug_bwd_split(flag)
        try:
            _lib.check(_lib.lib().tfa_bwd(C.byref(pb), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
        finally:
            _lib.debug_bwd_split(0)
        res.append((dq, dk, dv, delta))
    ref_delta = (dout.double() * out.double()).sum(-1)
    for dq, dk, dv, delta in res:
        assert bool(torch.isfinite(delta).all())
        assert (delta.double() - ref_delta).abs().max().item() <= 1e-5 * max(1.0, ref_delta.abs().max().item())
    assert (res[0][3] - res[1][3]).abs().max().item() <= 1e-5 * max(1.0, ref_delta.abs().max().item())
    for a, b in zip(res[0][:3], res[1][:3]):
        assert bool(((a.float() - b.float()).abs() <= 2.0 * ulp16(b.float(), dtype) + 1e-3 * b.float().abs().max()).all())   # (an ulp of the larger partial sums, where the result cancels)
    assert torch.equal(res[0][2], res[1][2])          # dV does not depend on delta


def test_bwd_deterministic_and_inputs_untouched(tfa, oracle, dev):
    from tiny_flash_attention_amd import ops

    q, k, v = (t.to(dev) for t in oracle.make_inputs(2, 4, 512, 128, torch.bfloat16, seed=9))
    dout = make_dout(2, 4, 512, 128, torch.bfloat16, 19).to(dev)
    keep = [t.clone() for t in (q, k, v, dout)]
    out, lse = ops.flash_attn_fwd(q, k, v, True, 0.088)
    g1 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, 0.088)
    g2 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, 0.088)
    torch.cuda.synchronize()
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)                      # no atomics: bit-reproducible
    for a, b in zip((q, k, v, dout), keep):
        assert torch.equal(a, b)


def test_bwd_autograd_function_bnhd(tfa, oracle, dev):
    # flash_attn_func is differentiable like the official function the reference compares against (test.py:71-76)
    q, k, v = oracle.make_inputs(2, 4, 300, 64, torch.float16, seed=4, Hk=2)
    qt, kt, vt = (t.transpose(1, 2).contiguous().to(dev).requires_grad_(True) for t in (q, k, v))   # (B,N,H,D)
    out = tfa.flash_attn_func(qt, kt, vt, causal=True)
    dout = make_dout(2, 4, 300, 64, torch.float16, 14)
    out.backward(dout.transpose(1, 2).contiguous().to(dev))
    ref = oracle.attn_bwd_reference(q, k, v, dout, True, 1.0 / 8.0)
    for g, r in zip((qt.grad, kt.grad, vt.grad), ref):
        assert (g.transpose(1, 2).double().cpu() - r).abs().max().item() <= 1e-2 * max(1.0, r.abs().max().item())


@pytest.mark.parametrize("mode", ["default", "workspace"])
@pytest.mark.parametrize("dtype,Nk", [(torch.float16, 200), (torch.float16, 333), (torch.bfloat16, 130)])
def test_bwd_padded_keys_with_strongly_negative_logits(tfa, oracle, dev, dtype, Nk, mode):
    """A ragged last key block is padded with zero K/V rows; with every logit strongly negative (LSE << -11) the padded keys'
    P = exp(0 - LSE) overflows 16 bits unless the kernel treats those keys as masked: inf in the kept dS turned into NaN in dQ
    through 0 * inf in the workspace form (round-3 advisor finding; the fused dK/dV launch now starts the padded lanes' S at -inf).
    Non-causal on purpose: the causal mask used to hide the padding."""
    from tiny_flash_attention_amd import ops

    B, H, Nq, D = 1, 2, 192, 64
    g = torch.Generator().manual_seed(77)
    q = (3.0 + torch.empty((B, H, Nq, D)).normal_(0, 0.5, generator=g)).to(dtype)
    k = (-3.0 + torch.empty((B, H, Nk, D)).normal_(0, 0.5, generator=g)).to(dtype)
    v = torch.empty((B, H, Nk, D)).normal_(0, 0.5, generator=g).to(dtype)
    dout = make_dout(B, H, Nq, D, dtype, 78)
    sc = 1.0 / math.sqrt(D)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, dout))
    out, lse = ops.flash_attn_fwd(qd, kd, vd, False, sc)
    assert lse.max().item() < -30.0                       # the regime of the finding
    ws = True if mode == "workspace" else None
    g32 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, False, sc, grad_f32=True, workspace=ws)
    g16 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, False, sc, workspace=ws)
    torch.cuda.synchronize()
    for name, t in zip(("dq", "dk", "dv"), g32):
        assert bool(torch.isfinite(t).all()), f"{name}: non-finite values ({mode} form)"
    check_grads(oracle, g32, g16, q, k, v, out.cpu(), dout, False, sc, dtype)


@pytest.mark.parametrize("mode", ["default", "workspace"])
def test_bwd_headline_whole_head_vs_oracle(tfa, oracle, dev, mode):
    """BASELINE config 3 at FULL size (B4 H32 N4096 D128 bf16 causal) through tfa_bwd, and one whole (b,h) head of dq, dk, dv —
    every row, every column — against the fp64 autograd oracle within the rigorous 16-bit rounding bounds (B1)-(B3).  (MHA: a head's
    gradients depend on that head's q, k, v, dout only, so the oracle differentiates the 4096 x 4096 problem of the one head.)"""
    from tiny_flash_attention_amd import ops

    B, H, N, D = 4, 32, 4096, 128
    g = torch.Generator(device=dev).manual_seed(31)
    mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    sc = 1.0 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(q, k, v, True, sc)
    ws = True if mode == "workspace" else None
    g32 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, sc, grad_f32=True, workspace=ws)
    g16 = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, sc, workspace=ws)
    torch.cuda.synchronize()
    b, h = 2, 13
    sl = lambda t: t[b:b + 1, h:h + 1].cpu()
    check_grads(oracle, tuple(sl(t) for t in g32), tuple(sl(t) for t in g16), sl(q), sl(k), sl(v), sl(out), sl(dout), True, sc, torch.bfloat16)


def test_bwd_headline_shape_properties(tfa, dev):
    """BASELINE config 3 at full size (B4 H32 N4096 D128 bf16 causal): finite, and head independence —
    permuting heads permutes every gradient bit-exactly; dV of row-constant dO equals column sums of P."""
    from tiny_flash_attention_amd import ops

    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda: torch.empty((2, 32, 4096, 128), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    sc = 1.0 / math.sqrt(128)
    out, lse = ops.flash_attn_fwd(q, k, v, True, sc)
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, sc)
    assert all(bool(torch.isfinite(t.float()).all()) for t in (dq, dk, dv))
    perm = torch.randperm(32, device=dev)
    qp, kp, vp, dp = (t[:, perm].contiguous() for t in (q, k, v, dout))
    outp, lsep = ops.flash_attn_fwd(qp, kp, vp, True, sc)
    dqp, dkp, dvp = ops.flash_attn_bwd(qp, kp, vp, outp, lsep, dp, True, sc)
    assert torch.equal(dqp, dq[:, perm]) and torch.equal(dkp, dk[:, perm]) and torch.equal(dvp, dv[:, perm])
    # sum_i dQ_i . q_i == sum_j dK_j . k_j  (both equal sum_ij dS_ij S_ij / scale ... the scale-homogeneity identity)
    lhs = (dq.float() * q.float()).sum(dim=(2, 3))
    rhs = (dk.float() * k.float()).sum(dim=(2, 3))
    assert (lhs - rhs).abs().max().item() <= 2e-2 * max(1.0, lhs.abs().max().item())


@pytest.mark.parametrize("dtype,B,H,Hk,Nq,Nk,D,causal,layout", [
    (torch.bfloat16, 2, 4, 2, 700, 900, 128, True, "bhnd"),
    (torch.float16, 1, 4, 4, 513, 513, 64, True, "bnhd"),
    (torch.bfloat16, 1, 6, 2, 320, 200, 96, False, "bnhd"),
    (torch.bfloat16, 1, 4, 2, 600, 777, 256, True, "bnhd"),      # the 256-wide kernel: its three launches each have a windowed instantiation (round 5)
    (torch.float16, 2, 2, 2, 400, 400, 192, False, "bhnd"),
    (torch.bfloat16, 1, 2, 1, 257, 1000, 136, True, "bnhd"),
])
def test_bwd_windowed_instantiations_return_the_same_bits(tfa, oracle, dev, dtype, B, H, Hk, Nq, Nk, D, causal, layout):
    """The windowed instantiations of the dQ launch and of the fused dK/dV launch (what slices of 2 GiB and more get) forced onto
    ordinary inputs through the debug knob: bit-identical gradients."""
    from tiny_flash_attention_amd import _lib, ops

    q, k, v = oracle.make_inputs(B, H, Nq, D, dtype, seed=61, Hk=Hk, Nk=Nk)
    dout = make_dout(B, H, Nq, D, dtype, 62)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, dout))
    if layout == "bnhd":
        qd, kd, vd, dod = (t.transpose(1, 2).contiguous() for t in (qd, kd, vd, dod))
    sc = 1.0 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(qd, kd, vd, causal, sc, layout=layout)
    g0 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, sc, layout=layout)
    _lib.debug_bwd_split(2)
    try:
        g1 = ops.flash_attn_bwd(qd, kd, vd, out, lse, dod, causal, sc, layout=layout)
        torch.cuda.synchronize()
    finally:
        _lib.debug_bwd_split(0)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)


def test_bwd_head_slices_beyond_2_gib_at_head_dim_256(tfa, dev):
    """The same at head dim 256 (round 5: TFA_ERR_STRIDE before): (B,N,H,D) storage, 256 heads of 256 — 128 KiB per row, 2.2 GiB per head slice; the
    256-wide kernel's three launches in their windowed instantiations, two heads against fp32 autograd on the device."""
    from tiny_flash_attention_amd import ops

    B, N, H, D = 1, 17000, 256, 256
    assert (N - 1) * H * D * 2 > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(94)
    mk = lambda: torch.empty((B, N, H, D), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    sc = 1.0 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(q, k, v, True, sc, layout="bnhd")
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, sc, layout="bnhd")
    torch.cuda.synchronize()
    del out
    idx = torch.arange(N, device=dev)
    for h in (0, 255):
        qh, kh, vh = (t[0, :, h].float().detach().requires_grad_(True) for t in (q, k, v))
        s_ = (qh @ kh.t()) * sc
        s_ = s_.masked_fill(idx[None, :] > idx[:, None], float("-inf"))
        o = torch.softmax(s_, dim=-1) @ vh
        o.backward(dout[0, :, h].float())
        for name, got, want in (("dq", dq[0, :, h], qh.grad), ("dk", dk[0, :, h], kh.grad), ("dv", dv[0, :, h], vh.grad)):
            gt = got.float()
            assert bool(torch.isfinite(gt).all()), name
            d = (gt - want).abs()
            bar = 2e-2 * want.abs().max().item() + 1e-3
            assert d.max().item() <= bar, f"head {h} {name}: max|d| {d.max().item():.3e} > {bar:.3e}"
            assert d[-640:].max().item() <= bar                              # the tail of the sequence: offsets beyond 2 GiB
        del s_, o, qh, kh, vh


def test_bwd_head_slices_beyond_2_gib_in_bnhd_layout(tfa, dev):
    """(B,N,H,D) storage with many heads: one (b,h) slice of q, k, v, dout and of every gradient spans 2.28e9 bytes — more than a
    32-bit buffer offset reaches.  tfa_bwd then runs the windowed instantiations; three heads are checked against fp32 autograd
    on the device (the rows near the END of the sequence are the ones whose offsets exceed 2 GiB)."""
    from tiny_flash_attention_amd import ops

    B, N, H, D = 1, 17408, 512, 128
    assert (N - 1) * H * D * 2 > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(93)
    mk = lambda: torch.empty((B, N, H, D), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    sc = 1.0 / math.sqrt(D)
    out, lse = ops.flash_attn_fwd(q, k, v, True, sc, layout="bnhd")
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, out, lse, dout, True, sc, layout="bnhd")
    torch.cuda.synchronize()
    del out
    idx = torch.arange(N, device=dev)
    for h in (0, 257, 511):
        qh, kh, vh = (t[0, :, h].float().detach().requires_grad_(True) for t in (q, k, v))
        s_ = (qh @ kh.t()) * sc
        s_ = s_.masked_fill(idx[None, :] > idx[:, None], float("-inf"))
        o = torch.softmax(s_, dim=-1) @ vh
        o.backward(dout[0, :, h].float())
        for name, got, want in (("dq", dq[0, :, h], qh.grad), ("dk", dk[0, :, h], kh.grad), ("dv", dv[0, :, h], vh.grad)):
            gt = got.float()
            assert bool(torch.isfinite(gt).all()), name
            d = (gt - want).abs()
            bar = 2e-2 * want.abs().max().item() + 1e-3
            assert d.max().item() <= bar, f"head {h} {name}: max|d| {d.max().item():.3e} > {bar:.3e}"
            assert d[-640:].max().item() <= bar                              # the tail of the sequence: offsets beyond 2 GiB
        del s_, o, qh, kh, vh
