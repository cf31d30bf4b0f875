"""GPU tests of split-KV: partial forwards over key chunks (tfa_fwd with kv_offset /
nk_total, fp32 partial O + LSE) merged by tfa_merge must reproduce the one-pass forward.

Tolerances: partial O is fp32, so after the merge the 16-bit result may differ from the one-pass kernel's only by
the different rounding points of P (tile boundaries are the same, the running reference differs per chunk):
(S1) vs the fp64 oracle the same bars as the forward (T1: 1e-2 for 16-bit out; T5: LSE 1e-4, same +inf pattern);
(S2) vs the one-pass kernel: |d| <= 2 ulp16 of the result + 2^-8 * A (A = sum_j P|v|);
(S3) tfa_merge alone vs the oracle's merge on identical fp32 partials: 1e-6 relative.
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


def _avail(variants):
    from tiny_flash_attention_amd import _lib

    return [v for v in variants if isinstance(v, str) or v < 0 or _lib.variant_available(v)]


@pytest.mark.parametrize("variant", _avail([-1, 17, 27, 30, "native", "native-chunks"]))      # "native": tfa_fwd_splitkv, all chunks in one launch; "native-chunks": its one-launch-per-chunk route (what slices >= 2 GiB and head dims > 128 take), forced
@pytest.mark.parametrize("dtype,B,H,Hk,Nq,Nk,D,causal,splits", [
    (torch.bfloat16, 1, 4, 4, 512, 512, 128, True, 2),
    (torch.bfloat16, 2, 4, 2, 300, 1000, 128, True, 3),      # GQA, ragged, Nq < Nk
    (torch.float16, 1, 2, 2, 700, 256, 64, True, 4),         # chunks that many rows cannot see at all
    (torch.float16, 1, 2, 1, 333, 777, 64, False, 5),
    (torch.bfloat16, 1, 1, 1, 1, 2048, 128, True, 8),        # decode-like: one query row, 8 key chunks
    (torch.bfloat16, 1, 2, 2, 300, 900, 96, True, 3),        # head dims inside the 128- and 64-wide kernels
    (torch.float16, 2, 2, 1, 200, 500, 32, False, 2),
    (torch.bfloat16, 1, 2, 2, 200, 900, 256, True, 3),       # head dims above 128: one launch of the 256-wide kernel per chunk
    (torch.float16, 2, 4, 2, 130, 700, 160, False, 4),
    (torch.bfloat16, 1, 4, 1, 1, 1500, 192, True, 5),        # decode row, MQA, D = 192
])
def test_splitkv_matches_one_pass(oracle, dev, variant, dtype, B, H, Hk, Nq, Nk, D, causal, splits):
    from tiny_flash_attention_amd import _lib, ops

    q, k, v = oracle.make_inputs(B, H, Nq, D, dtype, seed=31, Hk=Hk, Nk=Nk)
    sc = 1.0 / math.sqrt(D)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    chunks = variant == "native-chunks"
    native = variant == "native" or chunks
    if D > 128 and not (native or variant == -1):
        pytest.skip("head dims above 128 have one kernel: forced variants do not apply")
    if not native and variant >= 0 and not _lib.variant_available(variant):
        pytest.skip(f"kernel variant {variant} is an A/B arm: not in the product build (make EXPERIMENTAL=1)")
    _lib.set_variant(-1 if native else variant)
    try:
        full, lse_full = ops.flash_attn_fwd(qd, kd, vd, causal, sc)
        if chunks:
            _lib.debug_set_flags(8192)
        out, lse = ops.flash_attn_fwd_splitkv(qd, kd, vd, causal, sc, splits=splits, native=native)
        torch.cuda.synchronize()
    finally:
        _lib.debug_set_flags(0)
        _lib.set_variant(-1)
    exact, lse_x = oracle.exact64(q, k, v, causal, sc, return_lse=True)
    A = oracle.abs_weighted(q, k, v, causal, sc)
    o = out.float().cpu()
    assert bool(torch.isfinite(o).all())
    assert (o - exact).abs().max().item() <= 1e-2                                                  # (S1)
    fin = torch.isfinite(lse_x)
    assert bool((torch.isinf(lse.cpu()) == ~fin).all())
    if fin.any():
        assert (lse.cpu()[fin] - lse_x[fin]).abs().max().item() <= 1e-4
    d = (o - full.float().cpu()).abs()
    assert bool((d <= 2 * ulp16(exact, dtype) + 2.0 ** -8 * A + 1e-6).all())                       # (S2)
    assert torch.equal(torch.isinf(lse), torch.isinf(lse_full))


def test_merge_kernel_vs_oracle(oracle, dev):
    from tiny_flash_attention_amd import ops

    for D in (128, 96, 24):
        _merge_case(oracle, dev, D)


def _merge_case(oracle, dev, D):
    from tiny_flash_attention_amd import ops

    g = torch.Generator().manual_seed(3)
    P, B, H, N = 5, 2, 3, 77
    o_parts = torch.empty((P, B, H, N, D)).normal_(0, 1, generator=g)
    lse_parts = torch.empty((P, B, H, N)).normal_(0, 3, generator=g)
    lse_parts[1, :, :, :10] = math.inf            # empty parts for some rows
    lse_parts[:, 0, 0, 5] = math.inf              # a row whose parts are all empty
    o_parts[:, 0, 0, 5] = 0
    ref, lref = oracle.merge_partials(o_parts, lse_parts)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        out, lse = ops.merge_partials(o_parts.to(dev), lse_parts.to(dev), dt)
        torch.cuda.synchronize()
        tol = 1e-6 if dt == torch.float32 else (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11)
        assert ((out.float().cpu() - ref).abs() <= tol * ref.abs() + 1e-6).all()                   # (S3)
        fin = torch.isfinite(lref)
        assert bool((torch.isinf(lse.cpu()) == ~fin).all()) and (lse.cpu()[fin] - lref[fin]).abs().max().item() <= 1e-5
        assert float(out[0, 0, 5].float().abs().max()) == 0.0


def test_partial_forward_semantics(oracle, dev):
    # one chunk in the middle of a causal sequence: rows above the chunk see nothing (O = 0, LSE = +inf)
    from tiny_flash_attention_amd import ops

    q, k, v = oracle.make_inputs(1, 2, 512, 128, torch.bfloat16, seed=8)
    lo, hi = 192, 384
    o32, lse = ops.flash_attn_fwd(q.to(dev), k[:, :, lo:hi].to(dev), v[:, :, lo:hi].to(dev), True, 0.088, out_f32=True, kv_offset=lo, nk_total=512)
    torch.cuda.synchronize()
    ref, lref = oracle.partial_attn(q, k[:, :, lo:hi], v[:, :, lo:hi], True, 0.088, lo, 512)
    assert bool(torch.isinf(lse[:, :, :lo]).all()) and float(o32[:, :, :lo].abs().max()) == 0.0
    assert bool((torch.isinf(lse.cpu()) == torch.isinf(lref)).all())
    fin = torch.isfinite(lref)
    assert (lse.cpu()[fin] - lref[fin]).abs().max().item() <= 1e-4
    assert (o32.cpu() - ref).abs().max().item() <= 1e-2


def test_chunk_route_side_streams_and_graph_capture(oracle, dev):
    """Head dims above 128 (and slices of 2 GiB and more) run tfa_fwd_splitkv as one launch per key chunk, forked over the calling
    thread's side streams and joined before the merge.  (1) Same bits as the chunks launched in line (debug flag 16384);
    (2) the fork / join is legal inside a stream capture: the call is captured into a HIP graph and replayed on new data."""
    from tiny_flash_attention_amd import _lib, ops

    B, H, Hk, Nq, Nk, D = 1, 8, 4, 3, 6000, 256
    g = torch.Generator(device=dev).manual_seed(77)
    mk = lambda n, h: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(torch.bfloat16)
    q, k, v = mk(Nq, H), mk(Nk, Hk), mk(Nk, Hk)
    sc = 1.0 / math.sqrt(D)
    o1, l1 = ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=4)
    _lib.debug_set_flags(16384)
    try:
        o2, l2 = ops.flash_attn_fwd_splitkv(q, k, v, True, sc, splits=4)
    finally:
        _lib.debug_set_flags(0)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    full, lse_full = ops.flash_attn_fwd(q, k, v, True, sc)
    assert (o1.float() - full.float()).abs().max().item() <= 4e-3 and (l1 - lse_full).abs().max().item() <= 1e-4
    # graph capture and replay on fresh inputs
    sq, sk, sv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.flash_attn_fwd_splitkv(sq, sk, sv, True, sc, splits=4)      # warm-up outside the capture (lazy stream / attribute set-up)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        go, gl = ops.flash_attn_fwd_splitkv(sq, sk, sv, True, sc, splits=4)
    sq.copy_(q); sk.copy_(k); sv.copy_(v)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(go, o1) and torch.equal(gl, l1)
