"""ctypes binding of include/tfa.h.  Fails loudly when the HIP library is missing — there is
no CPU or PyTorch fallback on the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFA_LIB") or os.path.join(_HERE, "lib", "libtfa_hip.so")   # TFA_LIB: developer override (A/B of builds)

TFA_F16, TFA_BF16, TFA_F32 = 0, 1, 2

# every symbol include/tfa.h declares
SYMBOLS = (
    "tfa_version",
    "tfa_debug_mfma_ceiling",
    "tfa_strerror",
    "tfa_fwd",
    "tfa_fwd_bhnd",
    "tfa_fwd_bhnd_f32out",
    "tfa_fwd_plan",
    "tfa_fwd_variant",
    "tfa_fwd_rounding_rule",
    "tfa_fwd_time",
    "tfa_set_variant",
    "tfa_get_variant",
    "tfa_num_variants",
    "tfa_variant_name",
    "tfa_variant_available",
    "tfa_fwd_work",
    "tfa_debug_set_trace",
    "tfa_debug_set_flags",
    "tfa_debug_decode",
    "tfa_merge",
    "tfa_fwd_splitkv",
    "tfa_fwd_splitkv_workspace",
    "tfa_fwd_suggest_splits",
    "tfa_bwd",
    "tfa_bwd_plan",
    "tfa_debug_bwd_split",
    "tfa_bwd_workspace_bytes",
    "tfa_bwd_work",
    "tfa_bwd_time",
)


class TfaFwdParams(C.Structure):
    """struct tfa_fwd_params (include/tfa.h)."""

    _fields_ = [
        ("q", C.c_void_p),
        ("k", C.c_void_p),
        ("v", C.c_void_p),
        ("out", C.c_void_p),
        ("lse", C.c_void_p),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("Hk", C.c_int32),
        ("Nq", C.c_int32),
        ("Nk", C.c_int32),
        ("D", C.c_int32),
        ("q_stride", C.c_int64 * 3),
        ("k_stride", C.c_int64 * 3),
        ("v_stride", C.c_int64 * 3),
        ("o_stride", C.c_int64 * 3),
        ("softmax_scale", C.c_float),
        ("is_causal", C.c_int32),
        ("dtype", C.c_int32),
        ("out_dtype", C.c_int32),
        ("kv_offset", C.c_int64),
        ("nk_total", C.c_int64),
        ("flags", C.c_int32),
        ("reserved_", C.c_int32),
    ]


TFA_FWD_EXACT_MAX = 1   # tfa_fwd_params::flags (include/tfa.h)


class TfaBwdParams(C.Structure):
    """struct tfa_bwd_params (include/tfa.h)."""

    _fields_ = [
        ("q", C.c_void_p),
        ("k", C.c_void_p),
        ("v", C.c_void_p),
        ("out", C.c_void_p),
        ("dout", C.c_void_p),
        ("lse", C.c_void_p),
        ("dq", C.c_void_p),
        ("dk", C.c_void_p),
        ("dv", C.c_void_p),
        ("delta", C.c_void_p),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("Hk", C.c_int32),
        ("Nq", C.c_int32),
        ("Nk", C.c_int32),
        ("D", C.c_int32),
        ("q_stride", C.c_int64 * 3),
        ("k_stride", C.c_int64 * 3),
        ("v_stride", C.c_int64 * 3),
        ("o_stride", C.c_int64 * 3),
        ("do_stride", C.c_int64 * 3),
        ("dq_stride", C.c_int64 * 3),
        ("dk_stride", C.c_int64 * 3),
        ("dv_stride", C.c_int64 * 3),
        ("softmax_scale", C.c_float),
        ("is_causal", C.c_int32),
        ("dtype", C.c_int32),
        ("grad_dtype", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
    ]


class TfaError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"tfa status {status}: {text}")
        self.status = status


_lib = None


def lib():
    """Load (once) and return the C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tiny-flash-attention_amd/csrc`. There is no fallback path."
        )
    L = C.CDLL(LIB_PATH)
    P = C.POINTER(TfaFwdParams)
    L.tfa_version.restype = C.c_int
    L.tfa_strerror.restype = C.c_char_p
    L.tfa_strerror.argtypes = [C.c_int]
    L.tfa_fwd.restype = C.c_int
    L.tfa_fwd.argtypes = [P, C.c_void_p]
    bhnd = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.tfa_fwd_bhnd.restype = C.c_int
    L.tfa_fwd_bhnd.argtypes = bhnd
    L.tfa_fwd_bhnd_f32out.restype = C.c_int
    L.tfa_fwd_bhnd_f32out.argtypes = bhnd
    L.tfa_fwd_plan.restype = C.c_int
    L.tfa_fwd_plan.argtypes = [P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tfa_fwd_time.restype = C.c_int
    L.tfa_fwd_time.argtypes = [P, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    L.tfa_set_variant.restype = C.c_int
    L.tfa_set_variant.argtypes = [C.c_int]
    L.tfa_get_variant.restype = C.c_int
    L.tfa_fwd_variant.restype = C.c_int
    L.tfa_fwd_variant.argtypes = [P]
    L.tfa_fwd_rounding_rule.restype = C.c_int
    L.tfa_fwd_rounding_rule.argtypes = [P]
    L.tfa_num_variants.restype = C.c_int
    L.tfa_variant_name.restype = C.c_char_p
    L.tfa_variant_name.argtypes = [C.c_int]
    L.tfa_variant_available.restype = C.c_int
    L.tfa_variant_available.argtypes = [C.c_int]
    L.tfa_debug_set_flags.restype = C.c_int
    L.tfa_debug_set_flags.argtypes = [C.c_int]
    L.tfa_debug_set_trace.restype = C.c_int
    L.tfa_debug_set_trace.argtypes = [C.c_void_p]
    L.tfa_debug_mfma_ceiling.restype = C.c_int
    L.tfa_debug_mfma_ceiling.argtypes = [C.c_void_p, C.c_ulonglong, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
    L.tfa_fwd_splitkv.restype = C.c_int
    L.tfa_fwd_splitkv.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p]
    L.tfa_fwd_suggest_splits.restype = C.c_int
    L.tfa_fwd_suggest_splits.argtypes = [P]
    L.tfa_fwd_splitkv_workspace.restype = C.c_longlong
    L.tfa_fwd_splitkv_workspace.argtypes = [P, C.c_int]
    L.tfa_merge.restype = C.c_int
    L.tfa_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    PB = C.POINTER(TfaBwdParams)
    L.tfa_bwd.restype = C.c_int
    L.tfa_bwd.argtypes = [PB, C.c_void_p]
    L.tfa_bwd_plan.restype = C.c_int
    L.tfa_bwd_plan.argtypes = [PB]
    L.tfa_bwd_work.restype = C.c_int
    L.tfa_bwd_work.argtypes = [PB, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tfa_bwd_time.restype = C.c_int
    L.tfa_bwd_time.argtypes = [PB, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    L.tfa_fwd_work.restype = C.c_int
    L.tfa_fwd_work.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _lib = L
    return L


def check(status):
    if status != 0:
        raise TfaError(status, lib().tfa_strerror(status).decode())


def strerror(status):
    return lib().tfa_strerror(status).decode()


def set_variant(v):
    check(lib().tfa_set_variant(int(v)))


def debug_set_flags(flags):
    """Bring-up flags of the calling thread (include/tfa.h: tfa_debug_set_flags); 0 = normal."""
    check(lib().tfa_debug_set_flags(int(flags)))


def debug_bwd_split(on):
    """tfa_debug_bwd_split: dK and dV as two launches (A/B and cross-check of the fused dK/dV kernel)."""
    check(lib().tfa_debug_bwd_split(int(on)))       # bit 0: two-launch dK/dV form; bit 1: force the windowed (>= 2 GiB) instantiations


def get_variant():
    return lib().tfa_get_variant()


def num_variants():
    return lib().tfa_num_variants()


def variant_available(v):
    """True when kernel variant `v` is compiled into the loaded library (the product build carries only the
    dispatched kernels; A/B arms need `make EXPERIMENTAL=1`)."""
    return bool(lib().tfa_variant_available(int(v)))


def variant_name(v):
    return lib().tfa_variant_name(int(v)).decode()


def variant_for(B, H, Hk, Nq, Nk, D, is_causal, dtype=TFA_BF16, flags=0):
    """The variant tfa_fwd would run for a contiguous (B,H,N,D) problem of these sizes (no GPU needed)."""
    p = TfaFwdParams()
    p.q = p.k = p.v = p.out = 0x1000           # never dereferenced: tfa_fwd_variant only validates and plans
    p.lse = None
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, n, h in (("q_stride", Nq, H), ("k_stride", Nk, Hk), ("v_stride", Nk, Hk), ("o_stride", Nq, H)):
        arr = getattr(p, name)
        arr[0], arr[1], arr[2] = h * n * D, n * D, D
    p.softmax_scale = 1.0
    p.is_causal = 1 if is_causal else 0
    p.dtype = p.out_dtype = dtype
    p.flags = flags
    v = lib().tfa_fwd_variant(C.byref(p))
    if v < 0:
        check(v)
    return v


RULE_EXACT_MAX, RULE_LAZY, RULE_FIRST_TILE = 0, 1, 2


def rounding_rule(p):
    """The row reference tfa_fwd rounds P against for the problem *p (include/tfa.h: TFA_RULE_*)."""
    r = lib().tfa_fwd_rounding_rule(C.byref(p))
    if r < 0:
        check(r)
    return r


def rule_for(B, H, Hk, Nq, Nk, D, is_causal, dtype=TFA_BF16, flags=0):
    """rounding_rule for a contiguous (B,H,N,D) problem of these sizes (no GPU needed); honours a forced variant like variant_for."""
    p = TfaFwdParams()
    p.q = p.k = p.v = p.out = 0x1000
    p.lse = None
    p.B, p.H, p.Hk, p.Nq, p.Nk, p.D = B, H, Hk, Nq, Nk, D
    for name, n, h in (("q_stride", Nq, H), ("k_stride", Nk, Hk), ("v_stride", Nk, Hk), ("o_stride", Nq, H)):
        arr = getattr(p, name)
        arr[0], arr[1], arr[2] = h * n * D, n * D, D
    p.softmax_scale = 1.0
    p.is_causal = 1 if is_causal else 0
    p.dtype = p.out_dtype = dtype
    p.flags = flags
    return rounding_rule(p)


def lazy_reference(v):
    """True for the variants that keep a lazily re-based row reference instead of the exact running max (names "il..." / "x4...";
    "exact-il8", variant 38, is the il8 kernel with the exact running max)."""
    return variant_name(v).startswith("il") or variant_name(v).startswith("x4")
