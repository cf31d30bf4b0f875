"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own implementations (imported /
compiled from /root/reference) on the reference's fixture recipes, reduced in batch/heads so
the files stay small.  Run in the dev container only (the GPU box has no /root/reference):

    make -C oracle && python tests/golden/make_golden.py

Inputs are stored in the 16-bit dtype they were rounded to (so the GPU kernel can consume the
exact same bits); reference outputs are stored as fp32 (bf16/fp16 outputs upcast exactly).
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def as_np16(t):
    """bit pattern of a 16-bit tensor as uint16 (numpy has no bfloat16)."""
    return t.view(torch.int16).numpy().view(np.uint16)


def load_ref_py(name, path, stub_flash_attn=False):
    if stub_flash_attn:  # main_torch_only.py:4 imports the flash_attn pip package (absent here)
        m = types.ModuleType("flash_attn")
        m.flash_attn_func = None
        sys.modules.setdefault("flash_attn", m)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_tiny_py():
    """cfg1: flash_attention_py/tiny_flash_attn.py flash_attn_v2_multihead (:137-196) and the
    2-D flash_attn_v1/v2 (:4-135); no scale, no mask; B=1 H=2 N=128 D=64, BLOCK_M=4."""
    tfa = load_ref_py("tiny_flash_attn", f"{REF}/flash_attention_py/tiny_flash_attn.py")
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.empty((1, 2, 128, 64)).normal_(0.0, 0.5, generator=g).to(torch.float16)
    q, k, v = mk(), mk(), mk()
    out = tfa.flash_attn_v2_multihead(q.float(), k.float(), v.float(), "cpu", 4)
    out_v1 = tfa.flash_attn_v1(q[0, 0].float(), k[0, 0].float(), v[0, 0].float(), "cpu", 4)
    out_v2 = tfa.flash_attn_v2(q[0, 0].float(), k[0, 0].float(), v[0, 0].float(), "cpu", 4)
    np.savez_compressed(
        os.path.join(HERE, "tiny_py_cfg1.npz"),
        q=as_np16(q), k=as_np16(k), v=as_np16(v), dtype="float16",
        out_multihead=out.numpy(), out_v1_head0=out_v1.numpy(), out_v2_head0=out_v2.numpy(),
        scale=np.float32(1.0), causal=np.int32(0),
        source="flash_attention_py/tiny_flash_attn.py:137-196 flash_attn_v2_multihead(q,k,v,'cpu',4)",
    )


def golden_c_kernels():
    """flash_attention_c/test.py:34-48 recipe (seed 0, torch.rand, causal, scale 1/sqrt(D)),
    reduced to B=1 H=2; outputs of the reference's compiled naive_attn and flash_attn,
    plus a Nq != Nk case exercising the bottom-right causal offset (attn.cpp:121-124)."""
    from oracle import oracle as O

    ref = O.ref_kernels()
    assert ref is not None, "run `make -C oracle` first"
    torch.manual_seed(0)
    bs, hn, n, d = 1, 2, 128, 128
    q = torch.rand(bs, hn, n, d).to(torch.float16)
    k = torch.rand(bs, hn, n, d).to(torch.float16)
    v = torch.rand(bs, hn, n, d).to(torch.float16)
    sc = 1 / math.sqrt(d)
    res = {}
    for causal in (0, 1):
        res[f"naive_c{causal}"] = ref.naive_attn(q.float(), k.float(), v.float(), bool(causal), sc).numpy()
        res[f"flash_c{causal}"] = ref.flash_attn(q.float(), k.float(), v.float(), bool(causal), sc).numpy()
    # ragged: Nq=48 queries against Nk=128 keys (decode-style suffix), causal
    q2 = q[:, :, :48].contiguous()
    res["flash_nq48_c1"] = ref.flash_attn(q2.float(), k.float(), v.float(), True, sc).numpy()
    res["naive_nq48_c1"] = ref.naive_attn(q2.float(), k.float(), v.float(), True, sc).numpy()
    np.savez_compressed(
        os.path.join(HERE, "c_kernels_seed0.npz"),
        q=as_np16(q), k=as_np16(k), v=as_np16(v), dtype="float16", scale=np.float32(sc),
        source="flash_attention_c/csrc/attn.cpp naive_attn:206-234 flash_attn:237-262 (compiled unmodified)",
        **res,
    )


def golden_torch_only():
    """flash_attention_py/main_torch_only.py:281-290 recipe (seed 13, bf16, normal(0,0.5), causal,
    scale 1/sqrt(D), layout (B,N,H,D)), reduced to B=1 N=256 H=2 D=128; outputs of
    safe_self_attention (:9-42), flash_attention_v1 and flash_attention_v2 (:160-270)."""
    mto = load_ref_py("main_torch_only", f"{REF}/flash_attention_py/main_torch_only.py", stub_flash_attn=True)
    torch.manual_seed(13)
    B, N, H, D = 1, 256, 2, 128
    q, k, v = (t.to(torch.bfloat16) for t in mto.get_tensors(B, N, H, D))
    sc = 1 / math.sqrt(D)
    res = {}
    with torch.no_grad():
        for causal in (0, 1):
            res[f"safe_c{causal}"] = mto.safe_self_attention(q, k, v, is_causal=bool(causal), sm_scale=sc).float().numpy()
            res[f"v2_c{causal}"] = mto.flash_attention_v2(q, k, v, is_causal=bool(causal), sm_scale=sc).float().numpy()
        res["v1_c1"] = mto.flash_attention_v1(q, k, v, is_causal=True, sm_scale=sc).float().numpy()
    np.savez_compressed(
        os.path.join(HERE, "torch_only_seed13.npz"),
        q=as_np16(q), k=as_np16(k), v=as_np16(v), dtype="bfloat16", layout="bnhd", scale=np.float32(sc),
        source="flash_attention_py/main_torch_only.py safe_self_attention:9-42 flash_attention_v2:160-270",
        **res,
    )


if __name__ == "__main__":
    golden_tiny_py()
    golden_c_kernels()
    golden_torch_only()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
