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
    assert L.tfa_version() == 111


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
    big = dict(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128)
    _lib.set_variant(17)  # LDS-DMA kernel, two K and two V tile buffers, 128-row blocks paired (the split-KV kernel)
    st, grid, block, lds = plan(_params(**big))
    assert st == 0 and block == 256 and grid == 4 * 32 * 16 and lds == 4 * 64 * 128 * 2
    _lib.set_variant(-1)  # automatic: headline shape -> issue-interleaved kernel, 256-row blocks paired, 2 K + 2 V buffers
    st, grid, block, lds = plan(_params(**big))   # + one 32-row epilogue slice per wave + the max-free rule's "redo this pass" word (16 bytes) and one seed per query row
    assert st == 0 and block == 512 and grid == 4 * 32 * 8 and lds == 4 * 64 * 128 * 2 + 8 * 32 * 128 * 2 + 16 + 256 * 4
    # automatic, at most one 128-row block per CU (BASELINE config 2's shape, causal here) -> the key-split kernel: 8 waves per
    # 128-row block, unpaired, two groups of four tile buffers
    st, grid, block, lds = plan(_params(B=4, H=8, Hk=8, Nq=1024, Nk=1024, D=64, dtype=_lib.TFA_F16))
    assert st == 0 and block == 512 and grid == 4 * 8 * 8 and lds == 8 * 64 * 64 * 2 + 16 + 256 * 4
    # automatic, two 128-row blocks per CU -> 128-row blocks paired, two 4-wave workgroups per CU
    st, grid, block, lds = plan(_params(B=4, H=16, Hk=16, Nq=1024, Nk=1024, D=64, dtype=_lib.TFA_F16))
    assert st == 0 and block == 256 and grid == 4 * 16 * 4 and lds == 4 * 64 * 64 * 2 + 16 + 128 * 4
    if _lib.variant_available(1):   # A/B arms (make EXPERIMENTAL=1)
        _lib.set_variant(1)
        st, grid, block, lds = plan(_params(**big))
        assert st == 0 and block == 512 and grid == 4 * 32 * (4096 // 256) and lds == 4 * 64 * 128 * 2
        _lib.set_variant(2)
        st, grid, block, lds = plan(_params(B=4, H=8, Hk=8, Nq=1000, Nk=1000, D=64, dtype=_lib.TFA_F16))
        assert st == 0 and block == 256 and grid == 4 * 8 * 8 and lds == 4 * 64 * 64 * 2
        _lib.set_variant(11)  # LDS-DMA kernel: three K and three V tile buffers
        st, grid, block, lds = plan(_params(**big))
        assert st == 0 and block == 512 and grid == 4 * 32 * 8 and lds == 6 * 64 * 128 * 2
        _lib.set_variant(15)  # persistent: one workgroup per CU walks the 1024 work items
        st, grid, block, lds = plan(_params(**big))
        assert st == 0 and block == 512 and grid == 256
        _lib.set_variant(4)   # causal blocks paired: ceil(16/2) work items per head
        st, grid, block, lds = plan(_params(**big))
        assert st == 0 and block == 512 and grid == 4 * 32 * 8
    _lib.set_variant(-1)


def test_product_build_rejects_ab_arms():
    # the product library carries the dispatched kernels only (tfa_launch.h); the other table entries answer TFA_ERR_VARIANT
    L = _lib.lib()
    avail = [v for v in range(_lib.num_variants()) if _lib.variant_available(v)]
    for v in (17, 30, 32, 34, 36, 37, 38):
        assert v in avail
    assert len(avail) == 7 or len(avail) == _lib.num_variants()      # (make EXPERIMENTAL=1 builds carry every arm)
    for v in range(_lib.num_variants()):
        if v not in avail:
            assert L.tfa_set_variant(v) == -7
    assert _lib.get_variant() == -1


def test_rejects_bad_descriptors():
    cases = []
    p = _params(); p.q = None; cases.append((p, -1))
    cases.append((_params(dtype=2, out_dtype=1), -2))  # fp32 q,k,v (the correctness path) return fp32 only
    cases.append((_params(dtype=3), -2))               # no such dtype
    cases.append((_params(dtype=2, D=6), -3))          # fp32 path: head dims are multiples of 4
    cases.append((_params(out_dtype=0), -2))           # bf16 in, f16 out
    cases.append((_params(D=100), -3))                 # not a multiple of 8
    cases.append((_params(D=264), -3))                 # beyond the widest kernel
    st, grid, block, lds = plan(_params(dtype=2, B=2, H=4, Hk=2, Nq=100, Nk=333, D=72))   # fp32 tensors: one wave per 32 query rows, four waves per workgroup
    assert (st, grid, block, lds) == (0, (2 * 4 * 4 + 3) // 4, 256, 0)
    assert plan(_params(D=96))[0] == 0 and plan(_params(D=32, dtype=_lib.TFA_F16))[0] == 0 and plan(_params(D=8))[0] == 0
    st, grid, block, lds = plan(_params(B=2, H=4, Hk=4, Nq=1024, Nk=1024, D=256))   # x4-d256: 128-row blocks paired, five 32 KiB tile buffers
    assert st == 0 and block == 256 and grid == 2 * 4 * 4 and lds == 5 * 64 * 256 * 2
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


def test_slices_beyond_2_gib_are_windowed_not_refused():
    # (B,N,H,D) storage, H*D*2 = 128 KiB per row: a head slice of 20000 rows spans 2.6 GB.  The default il kernel takes it
    # (its windowed instantiation); the split-KV kernel (variant 17: one descriptor per slice) must refuse.
    p = _params(B=1, H=512, Hk=512, Nq=20000, Nk=20000, D=128)
    for name in ("q_stride", "k_stride", "v_stride", "o_stride"):
        a = getattr(p, name)
        a[0], a[1], a[2] = 20000 * 512 * 128, 128, 512 * 128
    assert plan(p)[0] == 0
    _lib.set_variant(17)
    try:
        assert plan(p)[0] == -5
    finally:
        _lib.set_variant(-1)
    p256 = _params(B=1, H=256, Hk=256, Nq=20000, Nk=20000, D=256)   # the 256-wide kernel has a windowed instantiation too (round 5)
    for name in ("q_stride", "k_stride", "v_stride", "o_stride"):
        a = getattr(p256, name)
        a[0], a[1], a[2] = 20000 * 256 * 256, 256, 256 * 256
    st, var = plan(p256)[0], _lib.variant_for(1, 256, 256, 20000, 20000, 256, False)
    assert st == 0 and _lib.variant_name(var).startswith("x4-d256")
    a = p.k_stride; a[2] = 8 * 1024 * 1024                  # 16 MiB per row: even one 64-row tile window exceeds 2 GiB
    assert plan(p)[0] == -5


def test_suggested_split_count():
    # decode-like: few query rows, long K/V, too few workgroups -> chunks; prefill and short caches -> one pass
    L = _lib.lib()
    sug = lambda **kw: L.tfa_fwd_suggest_splits(C.byref(_params(**kw)))
    assert sug(B=1, H=32, Hk=32, Nq=1, Nk=16384) == 8            # 32 workgroups -> 8 chunks fill 256 CUs
    assert sug(B=1, H=8, Hk=8, Nq=16, Nk=32768) == 32
    assert sug(B=1, H=8, Hk=8, Nq=16, Nk=5000) == 4              # at least 1024 keys per chunk
    assert sug(B=8, H=32, Hk=32, Nq=1, Nk=16384) == 1            # 256 workgroups already
    assert sug(B=4, H=32, Hk=32, Nq=1, Nk=16384) == 4            # half of the CUs: the split kernel fits two workgroups per CU
    assert sug(B=3, H=32, Hk=32, Nq=1, Nk=16384) == 5
    assert sug(B=5, H=32, Hk=32, Nq=1, Nk=16384) == 1            # 160 workgroups: a split no longer pays
    assert sug(B=1, H=32, Hk=32, Nq=1, Nk=2048) == 1             # short cache: the merge is not worth it
    assert sug(B=4, H=32, Hk=32, Nq=4096, Nk=4096) == 1
    assert sug(B=1, H=8, Hk=8, Nq=1, Nk=16384, D=256) == 16      # head dims above 128: the same LDS-DMA kernel, 256 wide


def test_split_suggestion_for_slices_beyond_one_descriptor():
    # a decode call over a long strided KV cache whose (b,h) slice spans more than 2 GiB: tfa_fwd plans it (windowed il kernel) and
    # tfa_fwd_splitkv takes its one-launch-per-chunk route through the same kernel (round 2 refused; ADVICE r02, medium: the
    # suggestion must never point at a call that fails) -> a small chunk count (four side streams carry the launches)
    L = _lib.lib()
    p = _params(B=1, H=8, Hk=1, Nq=1, Nk=300000, D=128)
    a = p.k_stride; a[0], a[1], a[2] = 300000 * 4096, 128, 4096
    a = p.v_stride; a[0], a[1], a[2] = 300000 * 4096, 128, 4096
    assert plan(p)[0] == 0
    assert L.tfa_fwd_suggest_splits(C.byref(p)) == 4
    small = _params(B=1, H=8, Hk=1, Nq=1, Nk=300000, D=128)        # the same problem with a dense cache: all chunks in one launch
    assert L.tfa_fwd_suggest_splits(C.byref(small)) > 4


def test_gqa_packing_is_an_optimisation_never_a_requirement():
    # q broadcast over the heads (head stride 0): as rows of a packed problem the "rows do not overlap" check fails, as the
    # caller described it the call is fine -> run() falls back to the caller's layout instead of returning TFA_ERR_STRIDE
    p = _params(B=64, H=32, Hk=8, Nq=1, Nk=8192)
    p.q_stride[1] = 0
    st, grid, _, _ = plan(p)
    assert st == 0 and grid == 64 * 32
    assert _lib.lib().tfa_fwd_variant(C.byref(p)) == _lib.lib().tfa_fwd_variant(C.byref(_params(B=64, H=32, Hk=32, Nq=1, Nk=8192)))


def test_rounding_rule_follows_dtype_variant_and_instantiation():
    """tfa_fwd_rounding_rule (include/tfa.h: TFA_RULE_*): bf16 on the main instantiation of an il kernel keeps the first key tile's row maximum
    (round 6, max-free); fp16, the special-case instantiations (a single partial query block: idle-wave form; head dims that leave the last column
    block empty: narrow form), head dims above 128 re-base lazily; the flag, the split-KV kernel and fp32 tensors follow the exact running maximum."""
    R = _lib.rule_for
    assert R(4, 32, 32, 4096, 4096, 128, True) == _lib.RULE_FIRST_TILE                              # the headline: il8, bf16
    assert R(4, 32, 32, 4096, 4096, 128, True, _lib.TFA_F16) == _lib.RULE_LAZY                      # fp16 P overflows at 2^16: keeps its maximum
    assert R(4, 8, 8, 1024, 1024, 64, False) == _lib.RULE_FIRST_TILE                                # config 2's shape in bf16: il4, 64 wide
    assert R(1, 8, 8, 4096, 4096, 128, True) == _lib.RULE_FIRST_TILE                                # the key-split kernel
    assert R(4, 32, 32, 4096, 4096, 96, True) == _lib.RULE_LAZY                                     # narrow instantiation (last 32-column block empty)
    assert R(64, 32, 32, 1, 8192, 128, True) == _lib.RULE_LAZY                                      # decode: the idle-wave instantiation
    assert R(4, 8, 8, 4096, 4096, 256, True) == _lib.RULE_LAZY                                      # the 256-wide kernel
    assert R(4, 32, 32, 4096, 4096, 128, True, flags=_lib.TFA_FWD_EXACT_MAX) == _lib.RULE_EXACT_MAX
    _lib.set_variant(17)
    try:
        assert R(4, 32, 32, 4096, 4096, 128, True) == _lib.RULE_EXACT_MAX                           # the burst-structured kernel, forced
    finally:
        _lib.set_variant(-1)
    assert _lib.lib().tfa_fwd_rounding_rule(None) == -1


def test_exact_max_flag_selects_the_exact_running_max_kernel():
    L = _lib.lib()
    p = _params(B=4, H=32, Hk=32, Nq=4096, Nk=4096)
    assert _lib.variant_name(L.tfa_fwd_variant(C.byref(p))).startswith("il8")
    p.flags = _lib.TFA_FWD_EXACT_MAX
    v = L.tfa_fwd_variant(C.byref(p))
    assert v == 38 and not _lib.lazy_reference(v)                  # the headline grid: the il8 kernel's exact-max instantiation (round 5)
    assert L.tfa_fwd_suggest_splits(C.byref(p)) == 1
    s = _params(B=1, H=4, Hk=4, Nq=1024, Nk=1024)                   # a small grid: the burst-structured LDS-DMA kernel, as before
    s.flags = _lib.TFA_FWD_EXACT_MAX
    assert L.tfa_fwd_variant(C.byref(s)) == 17 and not _lib.lazy_reference(17)
    g = _params(B=64, H=32, Hk=8, Nq=1, Nk=8192)
    g.flags = _lib.TFA_FWD_EXACT_MAX
    assert plan(g)[1] == 64 * 32                                   # no GQA row packing under the flag
    p.D = 256
    for name, (h, n) in (("q_stride", (32, 4096)), ("k_stride", (32, 4096)), ("v_stride", (32, 4096)), ("o_stride", (32, 4096))):
        a = getattr(p, name); a[0], a[1], a[2] = h * n * 256, n * 256, 256
    assert plan(p)[0] == -3                                         # no exact-max kernel above D = 128
    p.flags = 2
    assert plan(p)[0] == -4                                         # unknown flag bits are refused


def test_gqa_decode_is_planned_per_kv_head():
    # one query row, H/Hk = 4 query heads per K/V head: the library runs them as four rows of one problem (pack_gqa_rows) —
    # one workgroup per (batch, K/V head) instead of one per query head; a few causal rows likewise when the heads are adjacent
    def grid(**kw):
        p = _params(**kw)
        return plan(p)[1]
    assert grid(B=64, H=32, Hk=8, Nq=1, Nk=8192) == 64 * 8
    assert grid(B=64, H=32, Hk=32, Nq=1, Nk=8192) == 64 * 32
    assert grid(B=64, H=32, Hk=8, Nq=4, Nk=8192) == 64 * 8            # causal draft rows keep their positions (KArgs::row_mod)
    assert grid(B=64, H=32, Hk=8, Nq=64, Nk=8192) == 64 * 32           # 4 x 64 rows exceed one 128-row block: not packed
    _lib.set_variant(32)
    try:
        assert grid(B=64, H=32, Hk=8, Nq=1, Nk=8192) == 64 * 32       # a forced kernel variant runs the problem as given
    finally:
        _lib.set_variant(-1)


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


@pytest.mark.parametrize("module,names", [("attention_cutlass", ["flash_attention_v2_cutlass"]),
                                          ("attention_cuda", ["self_attention_cuda", "flash_attention_v1_cuda", "flash_attention_v2_cuda"]),
                                          ("_kernels", ["naive_attn", "flash_attn", "hello_world"])])
def test_reference_named_extension_modules_import_and_reject_cpu_tensors(module, names):
    """The three torch extension modules of the reference (flash_attention_cutlass/csrc/attention_api.cpp:6-10,
    flash_attention_cuda/csrc/attention_api.cpp:6-14, flash_attention_c/csrc/ops.cu:4-8) are importable under their own
    names from lib/, export the reference's function names, and refuse CPU tensors (there is no CPU path to fall into)."""
    import importlib
    import sys

    libdir = os.path.dirname(_lib.LIB_PATH)
    sys.path.insert(0, libdir)
    try:
        sys.modules.pop(module, None)
        mod = importlib.import_module(module)
    finally:
        sys.path.remove(libdir)
        sys.modules.pop(module, None)
    assert os.path.dirname(os.path.abspath(mod.__file__)) == os.path.abspath(libdir)
    q = torch.zeros(1, 1, 64, 64, dtype=torch.float16)
    for n in names:
        fn = getattr(mod, n)
        if n == "hello_world":
            continue
        args = (q, q, q) if module == "attention_cuda" else (q, q, q, True, 0.125)
        with pytest.raises(RuntimeError, match="q must be a CUDA tensor"):
            fn(*args)
        with pytest.raises(TypeError):
            fn(q, q)


def test_variant_selection_is_introspectable_without_gpu():
    # big grids -> the 8-wave issue-interleaved kernel, small grids -> its 4-wave form (128-row blocks, 2 workgroups per CU);
    # grids of at most one 128-row block per CU: causal, or non-causal from 4096 keys on -> the key-split form (BASELINE config 2: il4 since round 5)
    big = _lib.variant_for(4, 32, 32, 4096, 4096, 128, True)
    small = _lib.variant_for(4, 16, 16, 1024, 1024, 64, True, _lib.TFA_F16)
    cfg2 = _lib.variant_for(4, 8, 8, 1024, 1024, 64, False, _lib.TFA_F16)
    assert _lib.variant_name(big).startswith("il8-pair") and _lib.lazy_reference(big)
    assert _lib.variant_name(small).startswith("il4") and _lib.lazy_reference(small)
    assert _lib.variant_name(cfg2).startswith("il4") and _lib.lazy_reference(cfg2)
    assert _lib.variant_name(_lib.variant_for(1, 8, 8, 4096, 4096, 128, False)).startswith("il8-ksplit")
    assert _lib.variant_name(_lib.variant_for(4, 8, 8, 1024, 1024, 64, True, _lib.TFA_F16)).startswith("il8-ksplit")
    _lib.set_variant(17)
    try:
        assert _lib.variant_for(4, 32, 32, 4096, 4096, 128, True) == 17     # a forced variant is reported as such
    finally:
        _lib.set_variant(-1)
    with pytest.raises(_lib.TfaError):
        _lib.variant_for(1, 1, 1, 16, 16, 100, False)                        # unsupported head dim


def test_il_kernels_leave_the_pinned_accumulator_registers_alone():
    """tfa_fwd_kernel_il.h keeps O in v[192:255] by hand and caps the compiler at 192 VGPRs; if the cap is ever
    ignored (it was, once: LLVM doubles amdgpu-num-vgpr on gfx90a+), compiler-allocated code tramples O.  Disassemble
    the library: inside fwd_kernel_il* only v_mfma (acc in/out), 'v_mov_b32 vN, 0' and 'v_mul_f32' may touch v192+."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    if not (os.path.exists(objdump) and os.path.exists(bundler)):
        pytest.skip("ROCm LLVM tools not present")
    import tempfile

    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    dis = ""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        r = subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fat}", _lib.LIB_PATH], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            pytest.skip(f"cannot dump .hip_fatbin: {r.stderr[:200]}")
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        assert starts, "no offload bundles in .hip_fatbin"
        for i, a in enumerate(starts):                     # one bundle per translation unit
            b = starts[i + 1] if i + 1 < len(starts) else len(blob)
            part, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"b{i}.co")
            open(part, "wb").write(blob[a:b])
            r = subprocess.run([bundler, "--type=o", "--unbundle", f"--input={part}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            assert r.returncode == 0 and os.path.getsize(co) > 0, r.stderr[:300]
            dis += subprocess.run([objdump, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    cur, seen, bad = None, 0, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if not cur or ("fwd_kernel_il" not in cur and "bwd_kernel" not in cur):
            continue
        text = line.split("//")[0]
        # the il and backward kernels' LDS-DMA helper leaves M0 set (no save/restore): nothing else in them may touch M0
        if re.search(r"\bm0\b", text) and not text.split()[0:2] == ["s_mov_b32", "m0,"]:
            bad.append(f"{cur[:60]}: M0 used outside the LDS-DMA helper: {text.strip()}")
        if "fwd_kernel_il" not in cur:
            continue
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", text)]
        regs += [int(hi) for _, hi in re.findall(r"v\[(\d+):(\d+)\]", text)]
        if not regs or max(regs) < 192:
            continue
        seen += 1
        op = text.split()[0] if text.split() else ""
        ok = op.startswith("v_mfma_f32_32x32x16") or op == "v_mul_f32_e32" or (op == "v_mov_b32_e32" and text.rstrip().endswith(", 0"))
        if not ok:
            bad.append(f"{cur[:60]}: {text.strip()}")
    assert seen > 0, "no fwd_kernel_il code found in the library"
    assert not bad, "compiler-allocated use of the pinned O registers:\n" + "\n".join(bad[:10])


def _bwd_params(B=2, H=4, Hk=2, Nq=128, Nk=192, D=128, dtype=_lib.TFA_BF16, grad_dtype=None, base=0x10000):
    p = _lib.TfaBwdParams()
    for i, name in enumerate(("q", "k", "v", "out", "dout", "lse", "dq", "dk", "dv", "delta")):
        setattr(p, name, base * (i + 1))
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, (h, n) in (("q_stride", (H, Nq)), ("k_stride", (Hk, Nk)), ("v_stride", (Hk, Nk)), ("o_stride", (H, Nq)),
                         ("do_stride", (H, Nq)), ("dq_stride", (H, Nq)), ("dk_stride", (Hk, Nk)), ("dv_stride", (Hk, Nk))):
        a = getattr(p, name)
        a[0], a[1], a[2] = h * n * D, n * D, D
    p.softmax_scale = 0.1
    p.is_causal = 1
    p.dtype = dtype
    p.grad_dtype = dtype if grad_dtype is None else grad_dtype
    return p


def test_bwd_descriptor_validation_without_gpu():
    L = _lib.lib()
    assert L.tfa_bwd_plan(C.byref(_bwd_params())) == 0
    assert L.tfa_bwd_plan(C.byref(_bwd_params(grad_dtype=_lib.TFA_F32))) == 0          # fp32 gradients (debug path)
    assert L.tfa_bwd_plan(C.byref(_bwd_params(dtype=_lib.TFA_BF16, grad_dtype=_lib.TFA_F16))) == -2
    assert L.tfa_bwd_plan(C.byref(_bwd_params(D=96))) == 0                              # any multiple of 8 up to 256
    assert L.tfa_bwd_plan(C.byref(_bwd_params(D=192))) == 0 and L.tfa_bwd_plan(C.byref(_bwd_params(D=256))) == 0
    assert L.tfa_bwd_plan(C.byref(_bwd_params(D=100))) == -3 and L.tfa_bwd_plan(C.byref(_bwd_params(D=264))) == -3
    assert L.tfa_bwd_plan(C.byref(_bwd_params(H=4, Hk=3))) == -4
    p = _bwd_params(); p.delta = None
    assert L.tfa_bwd_plan(C.byref(p)) == -1
    p = _bwd_params(); p.dk_stride[2] = 64                                                # rows would overlap
    assert L.tfa_bwd_plan(C.byref(p)) == -5
    p = _bwd_params(); p.dout = 0x10008
    assert L.tfa_bwd_plan(C.byref(p)) == -6
    # (b,h) slices of 2 GiB and more ((B,N,H,D) storage, 128 KiB per row, 20000 rows): the windowed instantiations take them
    def strided(D):
        p = _bwd_params(B=1, H=512, Hk=512, Nq=20000, Nk=20000, D=D)
        for name in ("q_stride", "k_stride", "v_stride", "o_stride", "do_stride", "dq_stride", "dk_stride", "dv_stride"):
            a = getattr(p, name); a[0], a[1], a[2] = 20000 * 512 * D, D, 512 * D
        return p
    assert L.tfa_bwd_plan(C.byref(strided(128))) == 0
    assert L.tfa_bwd_plan(C.byref(strided(256))) == 0           # (round 5: the 256-wide launches have windowed instantiations too)
    huge = strided(256)
    huge.k_stride[2] = 8 * 1024 * 1024                          # 16 MiB per row: a 768-row window exceeds 2 GiB
    assert L.tfa_bwd_plan(C.byref(huge)) == -5
    # work model: 2.5 x the forward's flops (5 GEMMs), halved when causal
    f, b = C.c_double(), C.c_double()
    p = _bwd_params(B=4, H=32, Hk=32, Nq=4096, Nk=4096, D=128)
    assert L.tfa_bwd_work(C.byref(p), C.byref(f), C.byref(b)) == 0
    assert f.value == pytest.approx(2.5 * 5.498e11, rel=1e-3)
    assert b.value == pytest.approx(8 * 4 * 32 * 4096 * 128 * 2 + 4 * 4 * 32 * 4096, rel=1e-6)


def test_header_is_plain_c_and_links(tmp_path):
    """include/tfa.h must be usable from C (the boundary is a C ABI): compile a C translation unit that fills the
    descriptors and references every entry point, with gcc -std=c99 -Wall -Werror, and link it against the library."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not present")
    calls = "\n".join(f"  (void)&{name};" for name in header_symbols())
    src = tmp_path / "use_tfa.c"
    src.write_text(
        '#include "tfa.h"\n#include <string.h>\n'
        "int main(void) {\n"
        "  tfa_fwd_params p; tfa_bwd_params b;\n"
        "  memset(&p, 0, sizeof p); memset(&b, 0, sizeof b);\n"
        "  p.B = 1; p.H = 1; p.Hk = 1; p.Nq = 64; p.Nk = 64; p.D = 64; p.softmax_scale = 0.125f; p.dtype = TFA_BF16; p.out_dtype = TFA_BF16;\n"
        f"{calls}\n"
        "  return tfa_fwd_plan(&p, 0, 0, 0) == TFA_ERR_NULL && tfa_version() == TFA_VERSION ? 0 : 1;\n"
        "}\n")
    exe = tmp_path / "use_tfa"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe),
                        f"-L{libdir}", "-ltfa_hip", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, (run.returncode, run.stderr[:500])


def test_assembly_audit_tools_flag_what_they_exist_for(tmp_path):
    """The x4 and il units are built through two audits of hipcc's generated assembly (csrc/Makefile).  Feed them the two
    defects they were written for: an AGPR restore of an MFMA operand directly in front of an inline-asm MFMA (the x4
    kernel's row-block-0 bug of round 2) and a compiler-parked value in a hand-owned AGPR."""
    import subprocess
    import sys

    bad = tmp_path / "bad.s"
    bad.write_text(
        "kern:\n"
        "\tv_accvgpr_write_b32 a130, v7\n"                     # compiler parks a value in an owned register (outside asm)
        "\tds_read_b128 v[10:13], v1\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tv_accvgpr_read_b32 v13, a229\n"                     # VALU write of an MFMA operand ...
        "\t;;#ASMSTART\n"
        "\tv_mfma_f32_32x32x16_bf16 v[18:33], v[10:13], a[128:131], 0\n"   # ... zero wait states before the asm MFMA
        "\t;;#ASMEND\n"
        "\tv_add_f32_e32 v90, v18, v19\n"                      # and its result read before the passes are over
        "\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text(
        "kern:\n"
        "\tv_accvgpr_write_b32 a200, v7\n"
        "\tv_accvgpr_read_b32 v13, a229\n"
        "\ts_nop 1\n"
        "\t;;#ASMSTART\n"
        "\tv_mfma_f32_32x32x16_bf16 v[18:33], v[10:13], a[128:131], 0\n"
        "\t;;#ASMEND\n"
        "\t;;#ASMSTART\n"
        "\tv_mfma_f32_32x32x16_bf16 v[34:49], v[10:13], a[132:135], 0\n"
        "\t;;#ASMEND\n"
        "\t;;#ASMSTART\n"
        "\tv_mfma_f32_32x32x16_bf16 v[50:65], v[10:13], a[136:139], 0\n"
        "\t;;#ASMEND\n"
        "\t;;#ASMSTART\n"
        "\tv_mfma_f32_32x32x16_bf16 v[66:81], v[10:13], a[140:143], 0\n"
        "\t;;#ASMEND\n"
        "\tv_add_f32_e32 v90, v18, v19\n"                      # three MFMAs later: the result has left the pipe
        "\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n"
        "\ts_endpgm\n")
    hz = os.path.join(ROOT, "tools", "audit_mfma_hazard.py")
    ag = os.path.join(ROOT, "tools", "audit_agpr.sh")
    r = subprocess.run([sys.executable, hz, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "MFMA hazards: 2" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, hz, str(good)], capture_output=True, text=True)
    assert r.returncode == 0 and "MFMA hazards: 0" in r.stdout, r.stdout
    assert len(subprocess.run(["bash", ag, str(bad)], capture_output=True, text=True).stdout.strip().splitlines()) == 1
    assert subprocess.run(["bash", ag, str(good)], capture_output=True, text=True).stdout.strip() == ""


def test_loop_audit_models_the_in_order_lds_queue():
    """tools/audit_il_asm_loop.py: check() walks a disassembled statement.  Three things it exists for, on synthetic sequences: an MFMA that reads a fragment still in
    flight is reported; a store between two reads holds its place in the queue lgkmcnt counts (the backward's P hand-off: without it every later wait looks one short);
    a pack that writes an MFMA operand one instruction in front of the MFMA is reported."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_loop", os.path.join(ROOT, "tools", "audit_il_asm_loop.py"))
    au = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(au)
    P = lambda txt: [au.parse("\t" + l) for l in txt.strip().splitlines()]
    early = P("""
ds_read_b128 v[10:13], v1
ds_read_b128 v[14:17], v1 offset:64
s_waitcnt lgkmcnt(1)
v_mfma_f32_32x32x16_bf16 v[20:35], v[14:17], v[40:43], 0
""")
    assert any("[1]" in f for f in au.check(early)), au.check(early)
    with_store = P("""
ds_read_b128 v[10:13], v1
ds_write_b128 v2, v[60:63]
ds_read_b128 v[14:17], v1 offset:64
s_waitcnt lgkmcnt(2)
v_mfma_f32_32x32x16_bf16 v[20:35], v[10:13], v[40:43], 0
s_waitcnt lgkmcnt(0)
v_mfma_f32_32x32x16_bf16 v[20:35], v[14:17], v[40:43], v[20:35]
""")
    assert au.check(with_store) == [], au.check(with_store)
    short = P("""
v_cvt_pk_bf16_f32 v40, v50, v51
s_nop 0
v_mfma_f32_32x32x16_bf16 v[20:35], v[10:13], v[40:43], 0
""")
    assert any("[2]" in f for f in au.check(short)), au.check(short)


def test_work_item_decode_matches_plain_divisions():
    """the il kernels decode (b, h, k/v head, work item) from the workgroup id with host-computed magic-number divisions in one branch-free
    form (tfa_launch.h: fill_decode); tfa_debug_decode evaluates exactly that on the host.  Against the three dispatch orders written with
    plain divisions (GQA with (B * Hk) % 8 == 0: K/V heads round-robin over the XCDs; B * H % 8 == 0: heads round-robin; else (b,h)-major)."""
    import ctypes as C
    import itertools
    from tiny_flash_attention_amd import _lib

    L = _lib.lib()
    L.tfa_debug_decode.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)]
    out = (C.c_int * 4)()

    def plain(id, B, H, Hk, nwork):
        G = H // Hk
        if G > 1 and (B * Hk) % 8 == 0:
            x, s = id & 7, id >> 3
            per = G * nwork
            kg, r = x + 8 * (s // per), s % per
            bh, wi = (kg // Hk) * H + (kg % Hk) * G + r // nwork, r % nwork
        elif (B * H) % 8 == 0:
            x, s = id & 7, id >> 3
            bh, wi = x + 8 * (s // nwork), s % nwork
        else:
            bh, wi = id // nwork, id % nwork
        b, h = bh // H, bh % H
        return b, h, h // G, wi

    n = 0
    for B, H, Hk, nwork in itertools.product([1, 2, 3, 8, 13], [1, 2, 6, 8, 32, 40], [1, 2, 8], [1, 2, 3, 8, 17, 64]):
        if H % Hk:
            continue
        grid = B * H * nwork
        for id in list(range(min(grid, 300))) + [grid - 1, grid // 2]:
            assert L.tfa_debug_decode(B, H, Hk, nwork, id, out) == 0
            assert tuple(out) == plain(id, B, H, Hk, nwork), (B, H, Hk, nwork, id)
            n += 1
    # large operands: the magic numbers are exact for every dividend below 2^31
    for B, H, Hk, nwork, id in [(1, 1, 1, 1, 2**31 - 2), (64, 128, 8, 999, 64 * 128 * 999 - 1), (7, 33, 11, 12345, 7 * 33 * 12345 - 1), (8, 96, 8, 2731, 8 * 96 * 2731 - 5)]:
        assert L.tfa_debug_decode(B, H, Hk, nwork, id, out) == 0
        assert tuple(out) == plain(id, B, H, Hk, nwork)
    assert n > 10000


def test_outputs_that_alias_themselves_and_fp32_split_are_refused():
    """an `out` whose (b,h) slices share memory (broadcast batch / head stride, interleaved slices) would be written by several
    workgroups at once — TFA_ERR_STRIDE on the 16-bit and the fp32 path alike, while INPUTS may broadcast; fp32 q, k, v are tfa_fwd only:
    tfa_fwd_splitkv / its workspace query answer TFA_ERR_DTYPE on either of their routes."""
    L = _lib.lib()
    for dt in (_lib.TFA_BF16, _lib.TFA_F32):
        mk = lambda: _params(B=2, H=4, Hk=4, Nq=128, Nk=128, dtype=dt, out_dtype=_lib.TFA_F32 if dt == _lib.TFA_F32 else None)
        assert plan(mk())[0] == 0
        p = mk(); p.o_stride[1] = 0
        assert plan(p)[0] == -5, "out broadcast over heads"
        p = mk(); p.o_stride[0] = 0
        assert plan(p)[0] == -5, "out broadcast over the batch"
        p = mk(); p.o_stride[1] = 64 * 128                              # head slices overlap by half
        assert plan(p)[0] == -5
        p = mk(); p.q_stride[1] = 0; p.k_stride[0] = 0                  # inputs may broadcast
        assert plan(p)[0] == 0
        p = mk()                                                         # (B,N,H,D) output: heads interleaved inside a row — legal
        p.o_stride[0], p.o_stride[1], p.o_stride[2] = 128 * 4 * 128, 128, 4 * 128
        assert plan(p)[0] == 0
    f = _params(B=1, H=2, Hk=2, Nq=1, Nk=8192, dtype=_lib.TFA_F32, out_dtype=_lib.TFA_F32)
    L.tfa_fwd_splitkv_workspace.restype = C.c_longlong
    assert L.tfa_fwd_splitkv_workspace(C.byref(f), 4) == -2            # TFA_ERR_DTYPE
    assert L.tfa_fwd_splitkv(C.byref(f), 4, C.c_void_p(0x1000), None) == -2


def test_fp32_tensors_name_the_fix_on_the_16_bit_only_entries():
    """make_params / make_bwd_params raised a bare KeyError(torch.float32)."""
    import torch
    from tiny_flash_attention_amd import ops
    t = torch.zeros((1, 1, 8, 8), dtype=torch.float32)
    l = torch.zeros((1, 1, 8), dtype=torch.float32)
    with pytest.raises(TypeError, match="float16 or bfloat16 only"):
        ops.make_bwd_params(t, t, t, t, l, t, t, t, t, l, False, 1.0)


def test_generated_tile_loops_are_in_sync_with_their_generators():
    """csrc/tfa_fwd_il_asm_loop.inc, tfa_fwd_x4_asm_loop.inc and the backward's tfa_bwd_dq_asm_loop.inc / tfa_bwd_kv_asm_loop.inc are GENERATED (tools/gen_il_asm_loop.py,
    gen_x4_asm_loop.py, gen_bwd_dq_asm_loop.py, gen_bwd_kv_asm_loop.py): the committed text must be what
    the committed generators emit with their default knobs — an edit to either side without the other fails here, before it can reach a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("TFA_GEN_")}
    for gen, inc in (("gen_il_asm_loop.py", "tfa_fwd_il_asm_loop.inc"), ("gen_x4_asm_loop.py", "tfa_fwd_x4_asm_loop.inc"),
                     ("gen_bwd_dq_asm_loop.py", "tfa_bwd_dq_asm_loop.inc"), ("gen_bwd_kv_asm_loop.py", "tfa_bwd_kv_asm_loop.inc")):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", gen)], stdout=subprocess.PIPE, text=True, check=True, env=env).stdout
        have = open(os.path.join(root, "tiny-flash-attention_amd", "csrc", inc)).read()
        assert out == have, f"{inc} is stale: re-run tools/{gen}"


def test_generated_bodies_have_the_instruction_mix_the_docs_quote():
    """Round 6: the statement's bodies, counted from the generator itself (DESIGN.md section 2 / LABLOG L-13 quote these numbers): the max-free loop body carries
    115 VALU instructions per tile where the lazy one carries 130 (no v_max3 on S(j+1); a 3-instruction test of the row sums), every body its MFMAs — 32 in the
    pinned and the masked body, 24 / 8 in the half forms of the diagonal's even waves, 16 in the last-tile body — and the masked body two VALU per masked element."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_il", os.path.join(root, "tools", "gen_il_asm_loop.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)

    def mix(lines):
        ins = [l for l in lines if not (l.startswith(";") or l.startswith(".") or l.endswith(":"))]
        n = lambda pat: sum(1 for l in ins if re.match(pat, l))
        return {"all": len(ins), "mfma": n(r"v_mfma"), "valu": n(r"v_(?!mfma)"), "exp": n(r"v_exp"), "max": n(r"v_max"), "lds": n(r"ds_read"), "dma": n(r"buffer_load")}

    g.build("bf16", 128, 2)                                # sets the generator's shape globals; MAXFREE off
    lazy = mix(g.body(0, "t"))
    last, last_h = mix(g.last_body(0, "t")), mix(g.last_body(0, "t", half=True))
    masked, masked_h = mix(g.body(0, "t", tail=True, mask=True)), mix(g.body(0, "t", tail=True, mask=True, half=True))
    pinned = mix(g.body(0, "t", tail=True))
    assert lazy["all"] == 242 and lazy["mfma"] == 32 and lazy["valu"] == 130 and lazy["exp"] == 32 and lazy["max"] == 16 and lazy["lds"] == 48 and lazy["dma"] == 4, lazy
    assert pinned["mfma"] == 32 and pinned["valu"] == 128 and pinned["max"] == 16, pinned      # (no re-base test of its own: the dispatch behind it has it)
    assert masked["mfma"] == 32 and masked["valu"] - pinned["valu"] == 64 + 9, masked        # 32 elements x (v_cmp + v_cndmask) + the lane's limit and -inf (8) + thr again (1)
    assert masked_h["mfma"] == 24 and masked_h["lds"] == 40, masked_h                         # eight QK^T MFMAs and their fragment reads less
    assert last["mfma"] == 16 and last["valu"] == 112 and last["lds"] == 32 and last["max"] == 0, last
    assert last_h["mfma"] == 8 and last_h["valu"] == 56 and last_h["lds"] == 16, last_h
    lines_mf, _, n_mf, _, _ = g.build("bf16", 128, 2, maxfree=True)
    g.MAXFREE = True                                       # (build() leaves the switch off behind it: it writes the exact-running-max text last)
    mf = mix(g.body(0, "t"))
    assert n_mf == 227 and mf["all"] == 227 and mf["valu"] == 115 and mf["max"] == 2 and mf["mfma"] == 32, mf      # the two v_max are the test of the four partial row sums
    assert sum(1 for l in lines_mf if re.match(r"il_x[01]%=:", l)) == 2                        # leaving with a tile in hand forms its maximum on the way out
    g.build("bf16", 64, 1, maxfree=True)
    g.MAXFREE = True
    assert mix(g.body(0, "t"))["mfma"] == 16 and mix(g.body(0, "t"))["valu"] == 115


def test_backward_bodies_have_the_instruction_mix_the_docs_quote():
    """Round 6, the backward's two launches: a dQ tile is 48 MFMAs, 64 fragment reads, 6 LDS-DMA pieces and 80 + 32 VALU (scale/subtract, exp2, product, pack: 3.5 per
    score — dP starts at -delta, no subtraction); its masked body adds two VALU per score.  A dK/dV tile is 32 MFMAs per role: role 0 16 x (scale, subtract, exp2) + 8
    packs per half and the four P stores, role 1 16 x (subtract, unpack, product) + 8 packs and the four P loads; both read the tile's statistics as eight b128."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", name + ".py"))
        g = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(g)
        return g

    def mix(lines):
        ins = [l for l in lines if not (l.startswith(";") or l.startswith(".") or l.endswith(":"))]
        n = lambda pat: sum(1 for l in ins if re.match(pat, l))
        return {"mfma": n(r"v_mfma"), "valu": n(r"v_(?!mfma)"), "exp": n(r"v_exp"), "rd": n(r"ds_read"), "wr": n(r"ds_write"), "dma": n(r"buffer_load_dwordx4"), "bar": n(r"s_barrier")}

    dq = load("gen_bwd_dq_asm_loop")
    dq.build("bf16")
    m, mm = mix(dq.body(0)), mix(dq.body(0, mask=True))
    assert m == {"mfma": 48, "valu": 112, "exp": 32, "rd": 64, "wr": 0, "dma": 6, "bar": 1}, m
    assert mm["valu"] - m["valu"] == 64 + 1 and mm["mfma"] == 48, mm             # 32 scores x (v_cmp + v_cndmask) + the limit's step to the next tile
    kv = load("gen_bwd_kv_asm_loop")
    kv.build("bf16")
    r0, r1, r0m = mix(kv.body(0, 0)), mix(kv.body(1, 0)), mix(kv.body(0, 0, mask=True))
    assert r0["mfma"] == 32 and r0["valu"] == 112 + 3 and r0["exp"] == 32 and r0["rd"] == 48 + 8 and r0["wr"] == 4 and r0["dma"] == 4, r0   # (+ 3: lane * 4 for the statistics request)
    assert r1["mfma"] == 32 and r1["valu"] == 112 and r1["exp"] == 0 and r1["rd"] == 48 + 8 + 4 and r1["wr"] == 0 and r1["dma"] == 4, r1
    assert r0m["valu"] - r0["valu"] == 64 + 1, r0m
    kv.build("f16")                                        # fp16: the upper half of a P dword takes a shift in front of its conversion
    assert mix(kv.body(1, 0))["valu"] == 112 + 16


def test_library_carries_the_hand_scheduled_loops():
    """The product library must contain the generated steady-state loops (their asm labels survive as local symbols of the code objects): a build with
    -DTFA_IL_USE_ASMLOOP=0 / -DTFA_X4_USE_ASMLOOP=0 — the A/B arms — is not what ships."""
    raw = open(_lib.LIB_PATH, "rb").read()
    # one label per kernel that carries a loop: the lazy-reference loop in 8 units x (il8, il4, key split, key split paired) x two output types, the exact loop in
    # the 128-wide causal / non-causal units of both types, the 256-wide loop in 32 units
    # (round 6: the statement carries the bodies behind the loop — dispatch, masked, half and last-tile bodies — and, in the bf16 units, the max-free texts' exits)
    # (late round 6: the backward's dQ launch — 128 wide, causal x gradient type x dtype = 8 kernels — and the fused dK/dV launch, 8 kernels, with their masked bodies)
    for label, least in ((b"il_loop", 64), (b"ix_exit", 8), (b"x4_loop", 32), (b"il_mh0", 64), (b"il_lh1", 64), (b"il_d1", 64), (b"il_x0", 32), (b"ix_l0", 8),
                         (b"dq_loop", 8), (b"dq_m1", 8), (b"kv_r0m2", 8), (b"kv_r1b2", 8), (b"kv_exit", 8)):
        assert raw.count(label) >= least, f"{raw.count(label)} {label.decode()} labels in libtfa_hip.so, expected {least}: hand-scheduled loops are missing from this build"
    # ... and the windowed instantiations that (b,h) slices of 2 GiB and more run (round 5: the 256-wide forward kernel, VF | VF_X4_WINDOWED, and the 256-wide
    # backward kernel with BIG = true in all three modes)
    import re
    x4win = set(re.findall(rb"_ZN3tfa13fwd_kernel_x4I[0-9A-Za-z_]*Li1090519044E[0-9A-Za-z_]*", raw))
    assert len(x4win) == 8, sorted(x4win)                      # 2 dtypes x causal x output type
    bwdwin = set(re.findall(rb"_ZN3tfa10bwd_kernelI\w+?Li256ELi[012]ELb[01]ELb[01]ELb1ELi4ELb1ELi8E\w*", raw))
    assert len(bwdwin) == 24, len(bwdwin)                      # 2 dtypes x 3 modes x causal x gradient type
