"""The compressed 8-wide BVH of the big-scene ray kernels (mitsuba_amd/csrc/bvh.h: buildWide, k_wide.h), pinned on the CPU:
phip_debug_host_trace_wide walks the tree the builder emits with the SAME node-step arithmetic the device kernels compile
(wideNodeHits is __host__ __device__) and the same Wald test; compared with a brute-force sweep over every record and with
the oracle's kd-tree (the reference's structure).  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from mitsuba_amd import _abi as A, scene as S


def host_trace(phip, P, T, rays, wide=1):
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3); T = np.ascontiguousarray(T, np.uint32).reshape(-1, 3)
    r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    hits = np.zeros((len(r), 4), np.float32)
    info = A.phip_accel_info()
    rc = phip.phip_debug_host_trace_wide(P.ctypes.data_as(C.POINTER(C.c_float)), len(P), T.ctypes.data_as(C.POINTER(C.c_uint32)), len(T),
                                         r.ctypes.data_as(C.POINTER(A.phip_ray)), len(r), hits.ctypes.data_as(C.POINTER(A.phip_hit)), wide, C.byref(info), None, 0)
    assert rc == 0, phip.phip_last_error()
    return hits, info


def rays_through(rng, n, lo, hi, axis_aligned=0.0):
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = int(n * axis_aligned)
    if k:                                              # zero components (incl. -0.0): the slab test must not produce NaNs
        d[:k] = 0.0
        d[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
        d[: k // 2][d[: k // 2] == 0] = -0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf
    return rays


def soup(rng, n, size, extent):
    c = rng.uniform(-extent, extent, (n, 1, 3))
    P = (c + rng.normal(scale=size, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    return P, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


@pytest.mark.parametrize("n,size,extent", [(300, 1.0, 5.0), (5000, 0.3, 8.0), (40000, 0.05, 10.0), (2000, 4.0, 3.0)])
def test_wide_tree_equals_brute_force_on_triangle_soups(phip, n, size, extent):
    rng = np.random.default_rng(n)
    P, T = soup(rng, n, size, extent)
    rays = rays_through(rng, 3000, -extent, extent, axis_aligned=0.2)
    w, info = host_trace(phip, P, T, rays, 1)
    b, _ = host_trace(phip, P, T, rays, 0)
    assert info.n_nodes > 1 and info.node_bytes == 80
    assert info.n_triangle_refs == n if n < 4096 else n <= info.n_triangle_refs <= 1.5 * n + 64      # spatial splits (bvh.h) duplicate references from 4096 triangles on
    hit = b[:, 3].view(np.uint32) != A.PHIP_NO_HIT
    assert 0.05 < hit.mean() <= 1.0
    # structure-independent answer: (t, u, v, prim) bit for bit -- exact-t ties included (the highest triangle index wins, dv_scene.h: winsTie)
    assert (w.view(np.uint32) == b.view(np.uint32)).all()


def test_wide_tree_on_the_atrium_against_the_oracle_kd_tree(phip, oracle, gauss):
    """the Sponza-class scene of BASELINE.json configs[2]: long thin triangles, instanced columns, ~250k triangles"""
    sb = S.atrium(64, 36, gauss)
    desc = sb.desc()
    P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3)).copy()
    T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3)).copy()
    rng = np.random.default_rng(5)
    lo, hi = P.min(axis=0), P.max(axis=0)
    rays = rays_through(rng, 6000, lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo), axis_aligned=0.1)
    w, info = host_trace(phip, P, T, rays, 1)
    assert info.n_nodes < desc.n_triangles / 4 and info.max_depth <= 12
    osc = oracle.OracleScene(desc)
    oh, _, _ = osc.trace(rays, True, False)
    assert (oh[:, 3].view(np.uint32) != A.PHIP_NO_HIT).mean() > 0.9
    # the atrium has coincident coplanar surfaces (wall panels on the room shell): rays that hit both at the same t report
    # whichever the kd-tree tests last -- the distance is the same bit for bit, the primitive may differ ...
    same = (w.view(np.uint32) == oh.view(np.uint32)).all(axis=1)
    assert same.mean() >= 0.995, same.mean()
    assert (w[:, 0].view(np.uint32) == oh[:, 0].view(np.uint32)).all()
    # ... while the 8-wide tree (built with spatial splits here) returns the structure-independent answer: what a sweep over every
    # triangle in index order finds, ties included
    osc.set_bruteforce(True)                                   # (the kd-tree's ray-interval clipping and adaptive epsilon, then the sweep)
    ob, _, _ = osc.trace(rays, True, False)
    assert (w.view(np.uint32) == ob.view(np.uint32)).all()


def test_degenerate_inputs(phip):
    rng = np.random.default_rng(3)
    # coplanar triangles (one flat axis: the grid step of that axis is degenerate), duplicates, zero-area triangles
    n = 600
    P = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32); P[..., 1] = 0.25
    P[10] = P[11]; P[20, 1] = P[20, 0]                       # a duplicate and a zero-area triangle
    T = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    rays = rays_through(rng, 2000, -1.5, 1.5, axis_aligned=0.3)
    w, _ = host_trace(phip, P.reshape(-1, 3), T, rays, 1)
    b, _ = host_trace(phip, P.reshape(-1, 3), T, rays, 0)
    assert (w[:, 0].view(np.uint32) == b[:, 0].view(np.uint32)).mean() >= 0.999
    # identical centroids: the builder falls back to median splits and leaves of up to 8 records (split into pieces of 3)
    P = np.tile(rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32), (40, 1, 1))
    T = np.arange(120, dtype=np.uint32).reshape(40, 3)
    w, info = host_trace(phip, P.reshape(-1, 3), T, rays, 1)
    b, _ = host_trace(phip, P.reshape(-1, 3), T, rays, 0)
    assert (w[:, 0].view(np.uint32) == b[:, 0].view(np.uint32)).all()


def test_answer_does_not_depend_on_the_structure(phip, gauss, monkeypatch):
    """binned-SAH tree, spatial splits, cheaper / dearer traversal constants (different leaf contents and traversal orders), 8-wide
    tree and the plain sweep over the records: the same (t, u, v, prim) for every ray, bit for bit -- including the rays that hit the
    atrium's coincident wall panels, where two triangles report exactly the same distance (the highest index wins: winsTie; the node
    boxes are padded so that a tie at the current maxt is never culled: bvh.h pad())"""
    sb = S.atrium(64, 36, gauss, detail=0.5)
    desc = sb.desc()
    P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3)).copy()
    T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3)).copy()
    rng = np.random.default_rng(11)
    rays = rays_through(rng, 60000, P.min(axis=0), P.max(axis=0), axis_aligned=0.05)
    answers = []
    for env in ({"PHIP_BVH_SPATIAL": "0"}, {"PHIP_BVH_SPATIAL": "1"}, {"PHIP_BVH_SPATIAL": "1", "PHIP_BVH_CTRAV": "0.4", "PHIP_BVH_ALPHA": "1e-7"},
                {"PHIP_BVH_SPATIAL": "0", "PHIP_BVH_CTRAV": "2.5", "PHIP_BVH_MAXLEAF": "8"},
                {"PHIP_BVH_SPATIAL": "1", "PHIP_BVH_OPT": "2"}, {"PHIP_BVH_SPATIAL": "0", "PHIP_BVH_OPT": "1"}):     # + insertion-based re-optimisation of the topology
        for k in ("PHIP_BVH_SPATIAL", "PHIP_BVH_CTRAV", "PHIP_BVH_ALPHA", "PHIP_BVH_MAXLEAF", "PHIP_BVH_OPT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        w, info = host_trace(phip, P, T, rays, 1)
        answers.append((env, w.view(np.uint32).copy(), info.n_triangle_refs))
    b, _ = host_trace(phip, P, T, rays, 0)
    ties = 0
    for env, w, refs in answers:
        assert (w == answers[0][1]).all(), env
        assert (w == b.view(np.uint32)).all(), env
    assert answers[1][2] > answers[0][2] == desc.n_triangles          # the spatial builds did duplicate references


def test_big_degenerate_inputs_through_spatial_splits_and_reinsertion(phip):
    """inputs of >= 4096 triangles take the full builder (spatial splits + one pass of insertion-based re-optimisation): thousands of
    identical triangles, long slivers, and a completely flat scene of overlapping coplanar duplicates -- the build terminates with a
    shallow tree and the traversal returns what the sweep over all records returns, bit for bit -- also on the flat scene: the pad that
    keeps exact ties alive is taken from the scene's LARGEST extent on every axis (round 2 used the per-axis extent, which vanished
    for a scene flat in y, and which of twenty coincident copies was reported depended on the structure)"""
    rng = np.random.default_rng(1)
    tri = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32)
    flat = np.tile(rng.uniform(-1, 1, (300, 3, 3)).astype(np.float32) * np.array([1, 0, 1], np.float32), (20, 1, 1))
    slivers = (rng.uniform(-5, 5, (8000, 1, 3)) + rng.normal(size=(8000, 3, 3)) * np.array([3, 0.001, 0.001])).astype(np.float32)
    for name, P, exact in (("identical", np.tile(tri, (6000, 1, 1)), True), ("slivers", slivers, True), ("flat", flat, True)):
        n = len(P); T = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        rays = rays_through(rng, 2000, -2, 2, axis_aligned=0.2)
        w, info = host_trace(phip, P.reshape(-1, 3), T, rays, 1)
        b, _ = host_trace(phip, P.reshape(-1, 3), T, rays, 0)
        assert info.max_depth <= 16 and n <= info.n_triangle_refs <= 1.7 * n + 64, (name, info.max_depth, info.n_triangle_refs)
        assert (w[:, 0].view(np.uint32) == b[:, 0].view(np.uint32)).all(), name
        if exact:
            assert (w.view(np.uint32) == b.view(np.uint32)).all(), name


def test_build_time_is_bounded(phip):
    """scene creation pays for the builder (spatial splits + reinsertion from 4096 triangles on): the top levels' subtrees are built on
    several threads and the reinsertion search is bounded per insertion, so neither a 120 k-triangle soup nor 60 k coincident triangles
    (where the branch-and-bound of the reinsertion prunes nothing: O(n^2) without the bound) take long.  Generous limits: a regression
    of the kind round 2 shipped (10 s for 100 k coincident triangles, 121 s for a million) fails, machine noise does not."""
    rng = np.random.default_rng(5)
    n = 120000
    c = rng.uniform(-10, 10, (n, 1, 3)).astype(np.float32)
    soup = (c + rng.normal(0, 0.05, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    co = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (60000, 1))
    for name, P, limit_ms in (("soup", soup, 20000.0), ("coincident", co, 20000.0)):
        T = np.arange(len(P), dtype=np.uint32).reshape(-1, 3)
        info = A.phip_accel_info(); box = (C.c_float * 6)()
        rc = phip.phip_debug_host_build_bvh(np.ascontiguousarray(P).ctypes.data_as(C.POINTER(C.c_float)), len(P), T.ctypes.data_as(C.POINTER(C.c_uint32)), len(T), C.byref(info), box)
        assert rc == 0
        assert info.build_ms < limit_ms, (name, info.build_ms)
        assert info.max_depth <= 40, (name, info.max_depth)


def test_parallel_build_equals_the_serial_build(phip, monkeypatch):
    """the subtrees of the builder's top levels are built on several threads and spliced in depth-first order: same tree as one thread builds"""
    rng = np.random.default_rng(9)
    n = 40000
    c = rng.uniform(-4, 4, (n, 1, 3)).astype(np.float32)
    P = (c + rng.normal(0, 0.2, (n, 3, 3)).astype(np.float32) * np.array([1.0, 0.05, 0.3], np.float32)).reshape(-1, 3)
    T = np.arange(len(P), dtype=np.uint32).reshape(-1, 3)
    rays = rays_through(rng, 3000, -5, 5)
    res = []
    for threads in ("1", "8"):
        monkeypatch.setenv("PHIP_BVH_THREADS", threads)
        w, info = host_trace(phip, P, T, rays, 1)
        res.append((w.copy(), info.n_nodes, info.n_triangle_refs, info.max_depth, info.sah_cost))
    assert res[0][1:] == res[1][1:], (res[0][1:], res[1][1:])
    assert (res[0][0].view(np.uint32) == res[1][0].view(np.uint32)).all()
