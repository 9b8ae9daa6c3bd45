import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def phip():
    """libphip.so (built in-tree if needed).  Loads without a GPU; compute calls need one."""
    from mitsuba_amd import _ffi
    _ffi.build()
    return _ffi.lib()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.build()
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def gauss(oracle):
    """default reconstruction filter table (radius, table[32])"""
    return oracle.gaussian_filter(0.5)


@pytest.fixture(scope="session")
def have_gpu(phip):
    return phip.phip_device_count() > 0


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
