"""CPU tests: pin the oracle (oracle/ref_attn.c + oracle/oracle.py) to the reference.

(a) against golden vectors produced by the reference's own code (tests/golden/make_golden.py),
(b) against the reference's compiled CPU path (oracle/_ref) when it has been built,
(c) internal consistency of the three oracle formulations on edge cases."""
import math

import pytest
import torch

from helpers import load_golden


def test_golden_tiny_py_cfg1(oracle):
    # BASELINE config 1: tiny_flash_attn.py, B=1 H=2 N=128 D=64 fp32, scale 1, no mask
    z, dt, q, k, v = load_golden("tiny_py_cfg1.npz")
    want = torch.from_numpy(z["out_multihead"])
    for fn in (oracle.naive_attn, oracle.flash_attn, oracle.exact64):
        got = fn(q, k, v, False, 1.0)
        assert torch.allclose(got, want, rtol=0, atol=2e-6), fn.__name__
    # the restated Python double loop (what bench.py times as the "Python CPU path") reproduces the reference's own output
    assert torch.allclose(oracle.tiny_py_multihead(q, k, v, 4), want, rtol=0, atol=1e-6)
    # the reference's 2-D v1/v2 variants agree with head 0
    assert torch.allclose(torch.from_numpy(z["out_v1_head0"]), want[0, 0], atol=2e-6)
    assert torch.allclose(torch.from_numpy(z["out_v2_head0"]), want[0, 0], atol=2e-6)


@pytest.mark.parametrize("causal", [0, 1])
def test_golden_c_kernels(oracle, causal):
    z, dt, q, k, v = load_golden("c_kernels_seed0.npz")
    sc = float(z["scale"])
    assert torch.allclose(oracle.naive_attn(q, k, v, causal, sc), torch.from_numpy(z[f"naive_c{causal}"]), rtol=0, atol=1e-6)
    assert torch.allclose(oracle.flash_attn(q, k, v, causal, sc), torch.from_numpy(z[f"flash_c{causal}"]), rtol=0, atol=1e-6)
    assert torch.allclose(oracle.exact64(q, k, v, causal, sc), torch.from_numpy(z[f"flash_c{causal}"]), rtol=0, atol=2e-6)


def test_golden_c_kernels_ragged_causal(oracle):
    # Nq=48 < Nk=128: bottom-right aligned causal mask (attn.cpp:121-124)
    z, dt, q, k, v = load_golden("c_kernels_seed0.npz")
    sc = float(z["scale"])
    q2 = q[:, :, :48].contiguous()
    for fn, key in ((oracle.naive_attn, "naive_nq48_c1"), (oracle.flash_attn, "flash_nq48_c1"), (oracle.exact64, "flash_nq48_c1")):
        assert torch.allclose(fn(q2, k, v, True, sc), torch.from_numpy(z[key]), rtol=0, atol=2e-6)


@pytest.mark.parametrize("causal", [0, 1])
def test_golden_torch_only_bf16(oracle, causal):
    # main_torch_only.py: bf16 tensors in (B,N,H,D); reference tolerance atol=rtol=1e-2 (:309-312)
    z, dt, q, k, v = load_golden("torch_only_seed13.npz")
    sc = float(z["scale"])
    qb, kb, vb = (t.transpose(1, 2) for t in (q, k, v))  # -> (B,H,N,D)
    want_safe = torch.from_numpy(z[f"safe_c{causal}"]).transpose(1, 2)
    want_v2 = torch.from_numpy(z[f"v2_c{causal}"]).transpose(1, 2)
    exact = oracle.exact64(qb, kb, vb, causal, sc)
    torch.testing.assert_close(exact, want_safe, atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(exact, want_v2, atol=1e-2, rtol=1e-2)
    # our restatement of the tile loop reproduces the reference's tile loop output to bf16 rounding
    emu = oracle.tiled_emulation(qb.contiguous(), kb.contiguous(), vb.contiguous(), bool(causal), sc, 64)
    assert (emu.to(torch.bfloat16).float() - want_v2).abs().max() <= 2 ** -7  # <= 1 bf16 ulp at |o|<2


def test_against_compiled_reference_when_present(oracle):
    ref = oracle.ref_kernels()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    q, k, v = oracle.make_inputs(2, 3, 96, 64, torch.float32, seed=5)
    kk, vv = k[:, :, :80].contiguous(), v[:, :, :80].contiguous()
    for causal in (False, True):
        for (a, b, c) in ((q, k, v), (q[:, :, :40].contiguous(), kk, vv)):
            r_f = ref.flash_attn(a, b, c, causal, 0.2)
            r_n = ref.naive_attn(a, b, c, causal, 0.2)
            assert torch.allclose(oracle.flash_attn(a, b, c, causal, 0.2), r_f, rtol=0, atol=1e-6)
            assert torch.allclose(oracle.naive_attn(a, b, c, causal, 0.2), r_n, rtol=0, atol=1e-6)
            assert torch.allclose(oracle.exact64(a, b, c, causal, 0.2), r_f, rtol=0, atol=2e-6)


def test_oracle_formulations_agree_and_lse(oracle):
    q, k, v = oracle.make_inputs(1, 4, 160, 64, torch.float16, seed=3, Hk=2, Nk=200)  # GQA + Nq != Nk
    for causal in (False, True):
        o1, l1 = oracle.naive_attn(q, k, v, causal, 0.125, return_lse=True)
        o2, l2 = oracle.flash_attn(q, k, v, causal, 0.125, return_lse=True)
        o3, l3 = oracle.exact64(q, k, v, causal, 0.125, return_lse=True)
        o4, l4 = oracle.sdpa_reference(q, k, v, causal, 0.125)
        for o in (o1, o2, o4):
            assert torch.allclose(o, o3, rtol=0, atol=2e-6)
        for l in (l1, l2, l4):
            assert torch.allclose(l, l3, rtol=0, atol=2e-5)


def test_empty_rows(oracle):
    # causal with Nq > Nk: the first Nq-Nk rows see no key -> O = 0, LSE = +inf (flash_attention.cu:620-623)
    q, k, v = oracle.make_inputs(1, 1, 64, 64, torch.float16, seed=1, Nk=16)
    o, lse = oracle.exact64(q, k, v, True, 0.125, return_lse=True)
    assert torch.all(o[0, 0, :48] == 0) and torch.all(torch.isinf(lse[0, 0, :48]))
    assert torch.all(torch.isfinite(lse[0, 0, 48:]))


def test_rounding_helpers(oracle):
    L = oracle.lib()
    for x in (0.0, 1.0, 0.3, 1e-3, 0.999, 3.1415926, 1e-8):
        assert L.oracle_round_bf16(x) == torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).float().item()
        assert L.oracle_round_fp16(x) == torch.tensor(x, dtype=torch.float32).to(torch.float16).float().item()
    q, k, v = oracle.make_inputs(1, 1, 64, 64, torch.bfloat16, seed=2)
    a = oracle.exact64(q, k, v, False, 0.125)
    b = oracle.exact64(q, k, v, False, 0.125, p_round=torch.bfloat16)
    d = (a - b).abs().max().item()
    assert 0 < d < 2e-3  # rounding P moves the result, but only at the 2^-9 relative level


def test_tiled_emulation_general_shapes(oracle):
    # the tile-loop restatement agrees with the exact oracle up to the 16-bit rounding of P,
    # including GQA, Nq != Nk (both directions), ragged tiles and empty rows
    for (Nq, Nk, causal) in ((100, 333, False), (128, 384, True), (384, 128, True), (77, 77, True)):
        q, k, v = oracle.make_inputs(1, 4, Nq, 64, torch.bfloat16, seed=3, Hk=2, Nk=Nk)
        e, le = oracle.tiled_emulation(q, k, v, causal, 0.125, return_lse=True)
        x, lx = oracle.exact64(q, k, v, causal, 0.125, p_round=torch.bfloat16, return_lse=True)
        a = oracle.abs_weighted(q, k, v, causal, 0.125)
        assert bool(((e - x).abs() <= 2 ** -8 * a + 1e-6).all())      # worst-case P-rounding bound
        fin = torch.isfinite(lx)
        assert bool((torch.isfinite(le) == fin).all())
        assert (le[fin] - lx[fin]).abs().max().item() < 1e-5
        # with P left in fp32 the two formulations coincide to fp32 round-off
        e32 = oracle.tiled_emulation(q, k, v, causal, 0.125, p_dtype=torch.float32)
        assert (e32 - oracle.exact64(q, k, v, causal, 0.125)).abs().max().item() < 5e-6


@pytest.mark.parametrize("causal", [False, True])
def test_kernel_emulations_stay_on_the_exact_result(oracle, causal):
    """The emulations of the kernels' own rounding points (lazy re-base, key-split wave groups, GQA heads packed as rows with
    their query positions) are restatements of the same function: they must sit on exact64 within the P-rounding bound."""
    q, k, v = oracle.make_inputs(1, 4, 70, 64, torch.bfloat16, seed=9, Hk=2, Nk=333)
    sc = 0.125
    exact, lse_x = oracle.exact64(q, k, v, causal, sc, return_lse=True)
    A = oracle.abs_weighted(q, k, v, causal, sc)
    bound = 2.0 ** -8 * A + 1e-6
    for name, fn in (("lazy", lambda: oracle.tiled_emulation_lazy(q, k, v, causal, sc, 64, return_lse=True)),
                     ("ksplit", lambda: oracle.ksplit_emulation(q, k, v, causal, sc, 64, return_lse=True))):
        o, l = fn()
        assert bool(((o - exact).abs() <= bound).all()), name
        fin = torch.isfinite(lse_x)
        assert bool((torch.isinf(l) == ~fin).all()) and (l[fin] - lse_x[fin]).abs().max().item() <= 1e-4, name
    # GQA heads packed as rows: (B, Hk, G*Nq, D) with row position r % Nq (+ the causal shift Nk - Nq)
    q2, k2, v2 = oracle.make_inputs(2, 8, 5, 64, torch.float16, seed=10, Hk=2, Nk=200)
    ex2, lx2 = oracle.exact64(q2, k2, v2, causal, sc, return_lse=True)
    A2 = oracle.abs_weighted(q2, k2, v2, causal, sc)
    qp = q2.reshape(2, 2, 20, 64)
    kw = {"row_pos": torch.arange(20) % 5 + (200 - 5)} if causal else {}
    o2, l2 = oracle.tiled_emulation_lazy(qp, k2, v2, causal, sc, 64, return_lse=True, **kw)
    assert bool(((o2.reshape(ex2.shape) - ex2).abs() <= 2.0 ** -11 * A2 + 1e-6).all())
    assert (l2.reshape(lx2.shape) - lx2).abs().max().item() <= 1e-4


@pytest.mark.parametrize("causal", [False, True])
def test_first_tile_emulation_stays_on_the_exact_result(oracle, causal):
    """The max-free rule (round 6, include/tfa.h TFA_RULE_FIRST_TILE: P rounded against the row maximum of the FIRST key tile; power-of-two re-bases of the
    row sums; a query block whose row sum passes 2^64 redone with the lazy rule) restates the same function: pinned to exact64 inside the P-rounding
    bound on plain data, on scores that climb past 2^40 (the re-base) and past 2^64 / into fp32 overflow (the redo), plain and through the key split."""
    sc = 1.0 / math.sqrt(64)
    q, k, v = oracle.make_inputs(1, 2, 300, 64, torch.bfloat16, seed=11, Nk=333)
    cases = [("plain", q, k)]
    for name, spikes in (("rebase", ((5, 200, 14.0), (130, 330, 10.0))), ("redo", ((7, 150, 22.0), (290, 320, 60.0), (131, 331, 16.0)))):
        k2 = k.clone()
        for row, key, gain in spikes:                                       # a late key aligned with one query row: its score jumps by ~ gain * |q|^2 / 8
            k2[0, :, key] = (q[0, :, row].float() * gain).to(torch.bfloat16)
        cases.append((name, q, k2))
    for name, qq, kk in cases:
        exact, lse_x = oracle.exact64(qq, kk, v, causal, sc, return_lse=True)
        A = oracle.abs_weighted(qq, kk, v, causal, sc)
        fin = torch.isfinite(lse_x)
        o, l, redo = oracle.tiled_emulation_first_tile(qq, kk, v, causal, sc, 64, block_m=128, return_lse=True, return_redo=True)
        assert bool(redo.any()) == (name == "redo"), (name, redo)
        for tag, (o, l) in (("first_tile", oracle.tiled_emulation_first_tile(qq, kk, v, causal, sc, 64, block_m=128, return_lse=True)),
                            ("ksplit", oracle.ksplit_emulation(qq, kk, v, causal, sc, 64, return_lse=True, rule="first_tile"))):
            assert bool(torch.isfinite(o).all()), (name, tag)
            assert bool(((o - exact).abs() <= 2.0 ** -8 * A + 1e-6).all()), (name, tag, ((o - exact).abs() - 2.0 ** -8 * A).max().item())
            assert bool((torch.isinf(l) == ~fin).all()) and (l[fin] - lse_x[fin]).abs().max().item() <= 1e-4, (name, tag)
    # without a re-base or a redo the rule differs from the lazy one only where the lazy one re-bases: on plain data the two coincide bit for bit
    o_l = oracle.tiled_emulation_lazy(q, k, v, causal, sc, 64)
    o_f = oracle.tiled_emulation_first_tile(q, k, v, causal, sc, 64)
    assert bool((o_l == o_f).all())
