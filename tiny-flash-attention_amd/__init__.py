"""MI355X-native (gfx950) FlashAttention-2 forward — the one hot path of 66RING/tiny-flash-attention.

Host side: a thin Python mirror of the reference's operator interface over the C ABI of
``include/tfa.h`` (``lib/libtfa_hip.so``, hand-written HIP).  PyTorch is used only for device
memory, streams and torch.distributed.

Reference-named entry points (same names, argument order and return shapes):

* ``flash_attention_v2_cutlass(q, k, v, is_causal, softmax_scale) -> [out, lse]``
  (flash_attention_cutlass/csrc/attention_api.cpp:6-10, flash_attention.cu:741-772)
* ``flash_attention_v2_cuda(q, k, v) -> out`` (flash_attention_cuda/csrc/attention_api.cpp:6-14)
* ``flash_attn(q, k, v, is_causal, softmax_scale) -> out`` (flash_attention_c/csrc/ops.cu:4-8)
"""
from .ops import (  # noqa: F401
    flash_attention_v2_cutlass,
    flash_attention_v2_cuda,
    flash_attention_v1_cuda,
    self_attention_cuda,
    flash_attn,
    naive_attn,
    flash_attn_func,
    flash_attn_fwd,
    flash_attn_bwd,
    flash_attn_fwd_splitkv,
    merge_partials,
)
from . import _lib  # noqa: F401

__all__ = [
    "flash_attention_v2_cutlass",
    "flash_attention_v2_cuda",
    "flash_attention_v1_cuda",
    "self_attention_cuda",
    "flash_attn",
    "naive_attn",
    "flash_attn_func",
    "flash_attn_fwd",
    "flash_attn_bwd",
    "flash_attn_fwd_splitkv",
    "merge_partials",
]
