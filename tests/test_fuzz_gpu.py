"""Randomised cross-check of the dispatcher + every dispatched forward kernel against a device fp32 reference
(tools/fuzz_fwd.py): random shapes, head dims, GQA ratios, layouts, scales, causal or not — the cases nobody thought of."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["--n", "200", "--seed", "7"], ["--n", "80", "--seed", "8", "--big"], ["--n", "150", "--seed", "11", "--decode"],
                                  ["--n", "150", "--seed", "12", "--spec"]])
def test_fuzz_forward_against_device_fp32(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_fwd.py")] + args, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0, tail
    assert " ok; kernels used" in r.stdout


@pytest.mark.parametrize("args", [["--n", "120", "--seed", "9"],
                                  # the shapes that run the hand-scheduled statements of both backward launches: 128 wide, several whole tiles, GQA / MQA, Nq != Nk, ragged lengths
                                  ["--n", "120", "--seed", "10", "--focus", "asm"]])
def test_fuzz_backward_against_device_autograd(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bwd.py")] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "\n".join((r.stdout + r.stderr).splitlines()[-12:])
