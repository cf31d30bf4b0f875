import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: tens of seconds of host fp64 oracle work (all 128 headline heads); part of the default -m gpu suite")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()  # fail early if not built
    return O
