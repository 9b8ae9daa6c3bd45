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
    # torch brings its own copy of the HIP runtime: when libphip.so (linked against /opt/rocm's) has touched the GPU first, torch's later initialisation finds
    # "No HIP GPUs" -- the tests that hand torch tensors to the library then depend on the order they run in.  Initialise torch's side first.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
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


def sobol_tables(width, height, dimensions=128):
    """(matrices32, vdc, vdc_inv, log_resolution) for `default_render_params(sobol=...)` out of tests/golden/sobol_tables.npz (the reference
    plugin's direction numbers, tests/golden/make_sobol_tables.py) for a film whose crop window is width x height (sobol.cpp:147-157)"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "sobol_tables.npz"))
    m = int(np.ceil(np.log2(max(width, height, 1))))
    mat = np.ascontiguousarray(z["matrices32"][:dimensions * 52])
    if m > 1:
        return mat, np.ascontiguousarray(z["vdc"][m - 1]), np.ascontiguousarray(z["vdc_inv"][m - 1]), m
    return mat, np.zeros(52, np.uint64), np.zeros(52, np.uint64), m


def qmc_tables(scramble=-1, dimensions=64):
    """(primes, permutations or None) for `default_render_params(qmc=...)` out of tests/golden/qmc_tables.npz (the reference's prime table and the
    permutations of its PermutationStorage: Faure's for scramble -1, the pseudorandom ones for scramble 7, none for 0)"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "qmc_tables.npz"))
    primes = np.ascontiguousarray(z["primes"][:dimensions])
    if scramble == 0:
        return primes, None
    return primes, np.ascontiguousarray(z[{-1: "faure", 7: "random7"}[scramble]][:int(primes.sum())])
