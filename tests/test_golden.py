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


# ---- the reference's own renders (tests/golden/ref_renders.npz, written by tests/golden/make_golden_ref.py from
#      oracle/_ref = /root/reference compiled in place): the oracle has to reproduce them bit for bit ----
import pytest                                            # noqa: E402
import ref_scenes as RS                                  # noqa: E402


def _golden_mip(fixture, scene):
    def mip(key, image, kind, **kw):
        levels = []
        while "mip/%s/%s/%d" % (scene, key, len(levels)) in fixture:
            levels.append(np.ascontiguousarray(fixture["mip/%s/%s/%d" % (scene, key, len(levels))]))
        assert levels and np.array_equal(levels[0], image)      # level 0 = the (half-representable) source image
        return levels
    return mip


@pytest.mark.parametrize("name,build,kw", RS.CASES, ids=[c[0] for c in RS.CASES])
def test_reference_render_fixture(oracle, name, build, kw):
    """per-sample Li and the ImageBlock accumulator of the real MIPathTracer / MIDirectIntegrator (sampler `independent`)
    against the oracle's libm build on the same SFMT stream -- needs no reference tree"""
    fixture = np.load(os.path.join(G, "ref_renders.npz"))
    oracle.build(libm=True)
    gauss = oracle.gaussian_filter(0.5, libm=True)
    desc = build(gauss, _golden_mip(fixture, build.__name__)).desc()
    osc = oracle.OracleScene(desc, libm=True)
    film, samples, _ = osc.render(RS.params(kw), threads=1, sampler="sfmt", want_samples=True)
    ref_s, ref_f = fixture[name + "/samples"], fixture[name + "/film"]
    same = (samples.view(np.uint32) == ref_s.view(np.uint32)).all(-1).mean()
    assert same == 1.0, "%.4f%% of the samples are bit-identical to the reference (max abs diff %.3e)" % (100 * same, np.abs(samples - ref_s).max())
    assert np.array_equal(film.view(np.uint32), ref_f.view(np.uint32))


def test_halton_and_hammersley_known_answers(oracle):
    """The reference's own known answers for its radical-inverse samplers (src/tests/test_samplers.cpp:33-77: MATLAB's haltonset(5), the first five
    points; Hammersley with sampleCount = 5: i / 5 in front of the same columns), to the test's own tolerance of 1e-7.  They are the UNSCRAMBLED sequence
    (`scramble` = 0); the plugins' default, Faure's permutations, is pinned on the plugins themselves in tests/test_ref_pin.py."""
    import ctypes as C
    import numpy as np
    from conftest import qmc_tables
    L = oracle.lib()
    primes, perm = qmc_tables(0)
    assert perm is None and list(primes[:5]) == [2, 3, 5, 7, 11]
    pp = primes.ctypes.data_as(C.POINTER(C.c_uint32))
    halton = np.array([[0, 0, 0, 0, 0],
                       [0.500000000000000, 0.333333333333333, 0.200000000000000, 0.142857142857143, 0.090909090909091],
                       [0.250000000000000, 0.666666666666667, 0.400000000000000, 0.285714285714286, 0.181818181818182],
                       [0.750000000000000, 0.111111111111111, 0.600000000000000, 0.428571428571429, 0.272727272727273],
                       [0.125000000000000, 0.444444444444444, 0.800000000000000, 0.571428571428571, 0.363636363636364]])
    got = np.array([[L.oracle_rinv_sample(pp, len(primes), None, 0, 1, i, j) for j in range(5)] for i in range(5)])
    assert np.abs(got - halton).max() <= 1e-7
    hamm = np.concatenate([np.arange(5).reshape(5, 1) / 5.0, halton], axis=1)
    got = np.array([[L.oracle_rinv_sample(pp, len(primes), None, 1, 5, i, j) for j in range(6)] for i in range(5)])
    assert np.abs(got - hamm).max() <= 1e-7
    # test03_radicalInverseIncr: the van der Corput sequence, exactly
    for i in range(20):
        assert L.oracle_rinv_sample(pp, len(primes), None, 0, 1, i, 0) == int(format(i, "b")[::-1], 2) / float(1 << max(i.bit_length(), 1)) or i == 0
