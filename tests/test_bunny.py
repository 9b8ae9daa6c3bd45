"""The reference's own ray-cast test asset -- data/tests/bunny.ply, the "bunny benchmark" of src/tests/test_kd.cpp:86-128 (random
chords through a bounding sphere against ShapeKDTree::rayIntersect) -- as a committed fixture (tests/golden/bunny.npz, written
by tests/golden/make_golden_bunny.py in the build container): the mesh and the REFERENCE's answers to 40 000 chords.

  CPU (-m "not gpu"): the oracle's kd-tree and the host twin of the compressed wide BVH against the reference's answers
  GPU (-m gpu):       phip_trace (closest hit + any hit, the kdbench workload) against the reference's answers, and the asset
                      through the reference's `obj` mesh-loader plugin -> Scene -> path_hip shim -> GPU (where oracle/_ref is built)
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_l2
from mitsuba_amd import _abi as A

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bunny.npz")


@pytest.fixture(scope="module")
def bunny():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_bunny as G
    z = np.load(GOLDEN)
    rays = G.chords(int(z["n_chords"]), int(z["chord_seed"]))
    return dict(V=z["V"], F=z["F"].astype(np.uint32), rays=rays, t=z["ref_t"], uv=z["ref_uv"], shape=z["ref_shape"].astype(np.int32),
                prim=z["ref_prim"], scene=G.bunny_scene)


def agree(t, prim, b, what, min_prim=0.9999):
    """t bit for bit; the primitive as well except on exact ties (a chord through a shared edge)"""
    hit = np.isfinite(b["t"])
    assert (np.isfinite(t) == hit).all(), what
    assert (t[hit].view(np.uint32) == b["t"][hit].view(np.uint32)).all(), what
    on_bunny = hit & (b["shape"] == 0)                     # shape 0 = the bunny mesh: its primIndex is the face index
    assert (prim[on_bunny] == b["prim"][on_bunny].astype(np.uint32)).mean() >= min_prim, what


def test_oracle_kd_tree_on_the_bunny_equals_the_reference(oracle, gauss, bunny):
    desc = bunny["scene"](bunny["V"], bunny["F"], gauss).desc()
    osc = oracle.OracleScene(desc)
    oh, oo, _ = osc.trace(bunny["rays"], True, True)
    agree(oh[:, 0], oh[:, 3].view(np.uint32), bunny, "oracle")
    assert (oh[:, 1:3].view(np.uint32) == bunny["uv"].view(np.uint32)).all(axis=1)[np.isfinite(bunny["t"])].mean() > 0.9999
    assert (oo.astype(bool) == np.isfinite(bunny["t"])).all()           # test_kd counts rayIntersect(ray): the any-hit query
    assert 0.55 < np.isfinite(bunny["t"]).mean() < 0.68                 # the benchmark's ~61 % intersection rate


def test_wide_bvh_host_twin_on_the_bunny_equals_the_reference(phip, gauss, bunny):
    from test_wide_bvh import host_trace
    desc = bunny["scene"](bunny["V"], bunny["F"], gauss).desc()
    P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3)).copy()
    T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3)).copy()
    w, info = host_trace(phip, P, T, bunny["rays"], 1)
    assert info.n_nodes > 1000
    agree(w[:, 0], w[:, 3].view(np.uint32), bunny, "wide BVH (host twin)")


@pytest.mark.gpu
def test_phip_trace_on_the_bunny_equals_the_reference(phip, gauss, bunny):
    """the kdbench / test_kd workload through the C ABI (big-scene ray kernels: compressed wide BVH)"""
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible")
    from mitsuba_amd.integrator import Scene
    gs = Scene(bunny["scene"](bunny["V"], bunny["F"], gauss).desc())
    assert gs.accel_info().node_bytes == 80
    gh, go, st = gs.rayIntersect(bunny["rays"], True, True)
    agree(gh[:, 0], gh[:, 3].view(np.uint32), bunny, "phip_trace")
    assert (gh[:, 1:3].view(np.uint32) == bunny["uv"].view(np.uint32)).all(axis=1)[np.isfinite(bunny["t"])].mean() > 0.9999
    assert (go.astype(bool) == np.isfinite(bunny["t"])).all()
    print("bunny chords: %.1f nodes + %.1f records per closest-hit ray, %.1f nodes per any-hit ray" %
          (st.closest_node_visits / len(gh), st.closest_triangle_tests / len(gh), st.shadow_node_visits / len(gh)))
    gs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("loader", ["obj", "ply"])
def test_bunny_through_the_reference_obj_loader_and_the_shim(phip, gauss, bunny, tmp_path, loader):
    """a real asset, loaded by the reference's own mesh-loader plugins (src/shapes/obj.cpp, src/shapes/ply.cpp), flattened by the path_hip shim and
    rendered on the GPU, against the same mesh handed to the library directly"""
    from oracle import ref_ffi
    if not ref_ffi.available() or not ref_ffi.have_shims() or not os.path.exists(os.path.join(os.path.dirname(ref_ffi.LIB), "plugins", loader + ".so")):
        pytest.skip("oracle/_ref (reference build + plugin shims + %s loader) is not present" % loader)
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible")
    from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
    V, F = bunny["V"], bunny["F"]
    obj = tmp_path / ("bunny." + loader)
    if loader == "obj":
        with open(obj, "w") as f:
            for v in V:
                f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
            for t in F:
                f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))
    else:
        with open(obj, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face %d\n"
                     "property list uchar int vertex_indices\nend_header\n" % (len(V), len(F))).encode())
            f.write(V.astype("<f4").tobytes())
            rec = np.zeros(len(F), dtype=[("n", "u1"), ("i", "<i4", 3)]); rec["n"] = 3; rec["i"] = F
            f.write(rec.tobytes())
    w, h, spp = 96, 96, 16
    full = bunny["scene"](V, F, gauss, w, h)                    # shapes: bunny, floor, light
    # the same scene with the bunny coming from the file: the description holds floor + light, the loader adds the mesh LAST
    # (Scene::getShapes() order), so build the direct version in that order too
    from mitsuba_amd import scene as S
    sb = S.SceneBuilder()
    bunny_mat = sb.diffuse((0.6, 0.55, 0.5))
    sb.quad((-0.3, 0.032, -0.3), (0.3, 0.032, -0.3), (0.3, 0.032, 0.3), (-0.3, 0.032, 0.3), sb.diffuse((0.4, 0.4, 0.45)), facing=(0, 1, 0))
    sb.quad((-0.2, 0.5, -0.2), (0.2, 0.5, -0.2), (0.2, 0.5, 0.2), (-0.2, 0.5, 0.2), sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=(18.0, 17.0, 15.0))
    sb.perspective((-0.05, 0.18, 0.32), (-0.017, 0.10, 0.0), (0, 1, 0), 40.0)
    sb.hdrfilm(w, h, gauss)
    partial = sb.desc()
    rs = ref_ffi.RefScene(partial, shape_files=[(loader, str(obj), bunny_mat)])
    p = A.default_render_params(spp=spp, max_depth=6)
    img, sec = rs.render_job(p, threads=2, plugin="path_hip")  # Mitsuba: obj.so -> Scene -> path_hip.so -> libphip.so -> GPU
    sb.mesh(V, F, bunny_mat)                                    # the direct version: same shapes, same order
    gs = Scene(sb.desc())
    film = HDRFilm(w, h)
    assert PathHIP(maxDepth=6).render(gs, film, spp)
    direct = film.develop()
    r = rel_l2(img, direct)
    print("bunny through %s.so + path_hip.so vs direct: rel L2 %.3e (%.2f s)" % (loader, r, sec))
    assert np.isfinite(img).all() and img.max() > 0
    assert r < 1e-4
    cpu, _ = rs.render_job(p, threads=8)                        # the reference's CPU path on the loaded asset
    assert abs(img.mean() - cpu.mean()) / cpu.mean() < 0.08
    rs.close(); gs.close()
    del full


def test_bunny_through_the_reference_ply_loader_equals_the_mesh(gauss, bunny, tmp_path):
    """data/tests/bunny.ply is a PLY file; the reference reads such assets with src/shapes/ply.cpp, whose parser is written against Boost.MPL's lambda
    machinery (oracle/ref_shims/boost/mpl/vector.hpp restates the forms it uses).  The fixture's mesh written as PLY -- ascii and binary_little_endian, as
    ply.cpp's callbacks expect it: float x y z per vertex, a uchar-counted int list per face -- and loaded by the reference's own plugin answers the 40 000
    chords of the benchmark exactly as the reference answered them on the original file (CPU only: ShapeKDTree::rayIntersect on the loaded TriMesh)."""
    from oracle import ref_ffi
    if not ref_ffi.available() or not os.path.exists(os.path.join(os.path.dirname(ref_ffi.LIB), "plugins", "ply.so")):
        pytest.skip("oracle/_ref (reference build with the ply loader) is not present")
    from mitsuba_amd import scene as S
    V, F = bunny["V"].astype("<f4"), bunny["F"].astype("<i4")
    head = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
    asc = tmp_path / "bunny_ascii.ply"
    with open(asc, "w") as f:
        f.write(head % ("ascii", len(V), len(F)))
        for v in V:
            f.write("%.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
        for t in F:
            f.write("3 %d %d %d\n" % (t[0], t[1], t[2]))
    binf = tmp_path / "bunny_binary.ply"
    with open(binf, "wb") as f:
        f.write((head % ("binary_little_endian", len(V), len(F))).encode())
        f.write(V.tobytes())
        rec = np.zeros(len(F), dtype=[("n", "u1"), ("i", "<i4", 3)]); rec["n"] = 3; rec["i"] = F
        f.write(rec.tobytes())
    for path in (asc, binf):
        # the benchmark's scene is the bunny alone (make_golden_bunny.bunny_scene adds nothing else that a chord could hit first? it does: shape ids are
        # checked below) -- here: an empty description + the file, so every hit is on the loaded mesh
        sb = S.SceneBuilder()
        mat = sb.diffuse((0.5, 0.5, 0.5))
        # (a scene without emitter gets the reference's default sunsky, a plugin this build does not have: a millimetre of light a kilometre away)
        sb.quad((1000, 1000, 1000), (1000.001, 1000, 1000), (1000.001, 1000.001, 1000), (1000, 1000.001, 1000), sb.diffuse((0, 0, 0)), radiance=(1.0, 1.0, 1.0))
        sb.perspective((-0.05, 0.18, 0.32), (-0.017, 0.10, 0.0), (0, 1, 0), 40.0)
        sb.hdrfilm(16, 16, gauss)
        rs = ref_ffi.RefScene(sb.desc(), shape_files=[("ply", str(path), mat)])
        h = rs.trace(bunny["rays"])
        hits = h[0] if isinstance(h, tuple) else h
        t = np.asarray(hits)[:, 0]
        on_bunny = np.isfinite(bunny["t"]) & (bunny["shape"] == 0)
        assert np.isfinite(t[on_bunny]).all(), path
        assert (t[on_bunny].view(np.uint32) == bunny["t"][on_bunny].view(np.uint32)).all(), path      # the same distances, bit for bit
        miss = ~np.isfinite(bunny["t"])
        assert (~np.isfinite(t[miss])).all(), path
        rs.close()
