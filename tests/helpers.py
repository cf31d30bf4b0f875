"""Shared helpers for the test-suite (tests only)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[str(z["dtype"])]

    def t16(a):
        return torch.from_numpy(a.view(np.int16).copy()).view(dt)

    return z, dt, t16(z["q"]), t16(z["k"]), t16(z["v"])


def ulp16(ref32, dtype):
    """size of one unit-in-the-last-place of `dtype` at the magnitude of ref32 (elementwise)."""
    mant = 8 if dtype == torch.bfloat16 else 11
    a = ref32.abs().clamp_min(2.0 ** -14 if dtype == torch.float16 else 2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(a)) - (mant - 1))
