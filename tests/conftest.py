"""pytest configuration: `gpu` marker + shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI load/export checks.
`-m gpu` runs on an MI355X: parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """The C-ABI library, compiled in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    from memex_amd import _lib
    _lib.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def oracle():
    from oracle.search_oracle import COracle
    return COracle()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
