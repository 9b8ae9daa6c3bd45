"""Regenerates tests/golden/sobol_tables.npz and qmc_tables.npz: the first 128 dimensions of the direction numbers (sobol::Matrices::matrices32) and the 26 rows
of the van-der-Corput/Sobol' pixel-enumeration matrices, read as DATA out of the reference's own sobol plugin compiled in place
(oracle/_ref/plugins/sobol.so = src/samplers/sobol.cpp + sobolseq.cpp; `make -C oracle -f Makefile.ref`).  The GPU box has no /root/reference;
tests that need the tables there read this fixture (tests/test_ref_pin.py checks it against the plugin whenever the plugin is present).

    python tests/golden/make_sobol_tables.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
DIMS = 128

if __name__ == "__main__":
    from oracle import ref_ffi
    ref_ffi.build()
    so = C.CDLL(os.path.join(os.path.dirname(ref_ffi.__file__), "_ref", "plugins", "sobol.so"))
    mat = np.frombuffer((C.c_uint32 * (1024 * 52)).in_dll(so, "_ZN5sobol8Matrices10matrices32E"), dtype=np.uint32)[:DIMS * 52].copy()
    vdc = np.frombuffer((C.c_uint64 * (26 * 52)).in_dll(so, "_ZN5sobol8Matrices18vdc_sobol_matricesE"), dtype=np.uint64).reshape(26, 52).copy()
    inv = np.frombuffer((C.c_uint64 * (26 * 52)).in_dll(so, "_ZN5sobol8Matrices22vdc_sobol_matrices_invE"), dtype=np.uint64).reshape(26, 52).copy()
    np.savez_compressed(os.path.join(HERE, "sobol_tables.npz"), matrices32=mat, vdc=vdc, vdc_inv=inv)
    print("wrote sobol_tables.npz:", mat.shape, vdc.shape, inv.shape)
    # the radical-inverse samplers (halton, hammersley): the first 64 primes of the reference's table and the digit permutations its own
    # PermutationStorage builds (src/samplers/faure.cpp) -- Faure's (scramble = -1, the samplers' default) and the pseudorandom ones of scramble = 7
    primes, faure = ref_ffi.qmc_tables(-1, 64)
    _, random7 = ref_ffi.qmc_tables(7, 64)
    np.savez_compressed(os.path.join(HERE, "qmc_tables.npz"), primes=primes, faure=faure, random7=random7)
    print("wrote qmc_tables.npz:", primes.shape, faure.shape, random7.shape)
