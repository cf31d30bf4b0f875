"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/tfa.h declares, and rejects bad descriptors with the documented status codes
(validation runs through tfa_fwd_plan, which never touches a GPU)."""
import ctypes as C
import os
import re

import pytest
import torch

import tiny_flash_attention_amd as tfa
from tiny_flash_attention_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "tfa.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"include/tfa.h declares {n} but libtfa_hip.so does not export it"
    assert sorted(_lib.SYMBOLS) == names
    assert L.tfa_version() == 100


def _params(B=2, H=4, Hk=4, Nq=128, Nk=128, D=128, dtype=_lib.TFA_BF16, out_dtype=None, scale=0.1, base=0x10000):
    p = _lib.TfaFwdParams()
    p.q, p.k, p.v, p.out, p.lse = base, base * 2, base * 3, base * 4, base * 5
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, (h, n) in (("q_stride", (H, Nq)), ("k_stride", (Hk, Nk)), ("v_stride", (Hk, Nk)), ("o_stride", (H, Nq))):
        a = getattr(p, name)
        a[0], a[1], a[2] = h * n * D, n * D, D
    p.softmax_scale = scale
    p.is_causal = 1
    p.dtype = dtype
    p.out_dtype = dtype if out_dtype is None else out_dtype
    return p


def plan(p):
    g, b, l = C.c_int(), C.c_int(), C.c_int()
    st = _lib.lib().tfa_fwd_plan(C.byref(p), C.byref(g), C.byref(b), C.byref(l))
    return st, g.value, b.value, l.value


def test_plan_geometry():
    _lib.set_variant(1)
    st, grid, block, lds = plan(_params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128))
    assert st == 0 and block == 512 and grid == 4 * 32 * (4096 // 256) and lds == 4 * 64 * 128 * 2
    _lib.set_variant(2)
    st, grid, block, lds = plan(_params(B=4, H=8, Hk=8, Nq=1000, Nk=1000, D=64, dtype=_lib.TFA_F16))
    assert st == 0 and block == 256 and grid == 4 * 8 * 8 and lds == 4 * 64 * 64 * 2
    _lib.set_variant(11)  # LDS-DMA kernel: three K and three V tile buffers
    st, grid, block, lds = plan(_params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128))
    assert st == 0 and block == 512 and grid == 4 * 32 * 8 and lds == 6 * 64 * 128 * 2
    _lib.set_variant(15)  # persistent: one workgroup per CU walks the 1024 work items
    st, grid, block, lds = plan(_params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128))
    assert st == 0 and block == 512 and grid == 256
    _lib.set_variant(-1)  # automatic: causal N=4096 -> 128-row blocks, two LDS buffers, paired
    st, grid, block, lds = plan(_params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128))
    assert st == 0 and block == 256 and grid == 4 * 32 * 16 and lds == 4 * 64 * 128 * 2
    _lib.set_variant(4)   # causal blocks paired: ceil(16/2) work items per head
    st, grid, block, lds = plan(_params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128))
    assert st == 0 and block == 512 and grid == 4 * 32 * 8
    _lib.set_variant(-1)


def test_rejects_bad_descriptors():
    cases = []
    p = _params(); p.q = None; cases.append((p, -1))
    cases.append((_params(dtype=2), -2))
    cases.append((_params(out_dtype=0), -2))           # bf16 in, f16 out
    cases.append((_params(D=96), -3))
    cases.append((_params(Nq=0), -4))
    cases.append((_params(H=6, Hk=4), -4))
    p = _params(); p.k_stride[2] = 100; cases.append((p, -5))   # row stride not 16-byte aligned
    p = _params(); p.v_stride[2] = 64; cases.append((p, -5))    # rows overlap (stride < D)
    p = _params(base=0x10008); cases.append((p, -6))
    cases.append((_params(scale=0.0), -8))
    cases.append((_params(scale=float("nan")), -8))
    for p, want in cases:
        st, *_ = plan(p)
        assert st == want, (st, want)
        assert _lib.strerror(st).startswith("tfa:")
    assert _lib.lib().tfa_set_variant(99) == -7
    assert _lib.strerror(0) == "success"


def test_f32_out_and_gqa_accepted():
    assert plan(_params(out_dtype=_lib.TFA_F32))[0] == 0   # fp32 debug output
    assert plan(_params(H=8, Hk=2))[0] == 0
    assert plan(_params(Nq=77, Nk=333))[0] == 0


def test_work_model():
    p = _params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128)
    f, b = C.c_double(), C.c_double()
    assert _lib.lib().tfa_fwd_work(C.byref(p), C.byref(f), C.byref(b)) == 0
    assert f.value == pytest.approx(5.498e11, rel=1e-3)
    assert b.value == pytest.approx(536.9e6 + 2.1e6, rel=1e-3)
    p.is_causal = 0
    _lib.lib().tfa_fwd_work(C.byref(p), C.byref(f), C.byref(b))
    assert f.value == pytest.approx(1.0995e12, rel=1e-3)


def test_operator_error_behaviour_without_gpu():
    # CHECK_INPUT semantics of the reference binding (attention_api.cuh:12-18): CPU tensors are rejected
    q = torch.zeros(1, 1, 64, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="q must be a CUDA tensor"):
        tfa.flash_attention_v2_cutlass(q, q, q, True, 0.125)
    with pytest.raises(RuntimeError, match="q must be a CUDA tensor"):
        tfa.flash_attention_v2_cuda(q, q, q)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        tfa.flash_attn(q, q, q, False, 1.0)
    # all five arguments are positional in the reference binding (no py::arg): too few -> TypeError
    with pytest.raises(TypeError):
        tfa.flash_attention_v2_cutlass(q, q, q)
