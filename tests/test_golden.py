"""The oracle against the reference's own golden vectors."""
import ctypes as C
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sfmt_seed4321_words(oracle):
    """src/tests/test_random.cpp:433-473 -- Random(4321) first 64-bit outputs"""
    g = json.load(open(os.path.join(G, "sfmt_seed4321.json")))
    n = len(g["words_hex"])
    assert n >= 100
    out = (C.c_uint64 * n)()
    oracle.lib().oracle_sfmt_words(g["seed"], n, out)
    assert [int(w, 16) for w in g["words_hex"]] == list(out)


def test_sfmt_crosses_state_refill(oracle):
    """more than one 312-word state block, determinism and the [0,1) float mapping of random.cpp:632-641"""
    L = oracle.lib()
    n = 1000
    a = (C.c_uint64 * n)(); b = (C.c_uint64 * n)()
    L.oracle_sfmt_words(5489, n, a); L.oracle_sfmt_words(5489, n, b)
    assert list(a) == list(b)
    assert len(set(a)) == n
    f = np.zeros(4096, np.float32)
    L.oracle_sfmt_floats(5489, 0, len(f), f.ctypes.data_as(C.POINTER(C.c_float)))
    assert (f >= 0).all() and (f < 1).all()
    w = np.array(list(a), dtype=np.uint64)
    expect = (((w & np.uint64(0xFFFFFFFF)) >> np.uint64(9)).astype(np.uint32) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1)
    assert (f[:n] == expect).all()
    assert abs(float(f.mean()) - 0.5) < 0.02


def test_sfmt_clone_streams_differ(oracle):
    """independent.cpp:71-80: worker samplers are re-seeded by init_by_array from the parent"""
    L = oracle.lib()
    f0 = np.zeros(64, np.float32); f1 = np.zeros(64, np.float32)
    L.oracle_sfmt_floats(5489, 0, 64, f0.ctypes.data_as(C.POINTER(C.c_float)))
    L.oracle_sfmt_floats(5489, 1, 64, f1.ctypes.data_as(C.POINTER(C.c_float)))
    assert not (f0 == f1).any()
    assert (f1 >= 0).all() and (f1 < 1).all()


def test_clipped_aabb_known_answers(oracle):
    """src/tests/test_kd.cpp:34-84 -- Triangle::getClippedAABB"""
    g = json.load(open(os.path.join(G, "clipped_aabb.json")))
    tri = (C.c_float * 9)(*g["triangle"])
    for case in g["cases"]:
        box = (C.c_float * 6)(*case["box"]); out = (C.c_float * 6)()
        valid = oracle.lib().oracle_clipped_aabb(tri, box, out)
        assert bool(valid) == case["valid"], case
        if case["valid"]:
            assert list(out)[:3] == case["min"] and list(out)[3:] == case["max"], (case, list(out))
