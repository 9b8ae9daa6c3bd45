#!/usr/bin/env python3
"""Writes tests/golden/bunny.npz: the reference's own ray-cast test asset, data/tests/bunny.ply (src/tests/test_kd.cpp:86-128,
the "bunny benchmark": random chords through a bounding sphere against ShapeKDTree::rayIntersect), as arrays, together with the
REFERENCE's answers to a fixed chord workload -- computed here, in the build container, by the reference itself
(oracle/_ref: Scene::rayIntersect on its SAH kd-tree).  The fixture travels to the GPU box, /root/reference does not.

    python tests/golden/make_golden_bunny.py

The reference's `ply` loader cannot be built in this image (Boost.MPL, oracle/Makefile.ref), so the file is parsed here
(binary little-endian, float x/y/z vertices, uchar/int face lists) and handed to the reference as a TriMesh."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("MTS_REFERENCE", "/root/reference")


def read_ply(path):
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii").split("\n")
    assert "format binary_little_endian 1.0" in header
    nv = nf = 0; vprops = []; section = None
    for line in header:
        t = line.split()
        if t[:2] == ["element", "vertex"]: nv = int(t[2]); section = "v"
        elif t[:2] == ["element", "face"]: nf = int(t[2]); section = "f"
        elif t and t[0] == "property" and section == "v": vprops.append((t[2], t[1]))
        elif t and t[0] == "property" and section == "f": assert t[1:4] == ["list", "uchar", "int"], line
    assert [p[0] for p in vprops[:3]] == ["x", "y", "z"] and all(p[1] == "float" for p in vprops)
    V = np.frombuffer(raw, "<f4", nv * len(vprops), end).reshape(nv, len(vprops))[:, :3].copy()
    off = end + nv * len(vprops) * 4
    F = np.frombuffer(raw, np.dtype([("n", "u1"), ("i", "<i4", 3)]), nf, off)
    assert (F["n"] == 3).all()
    return V, F["i"].astype(np.uint32).copy()


def chords(n, seed=1234):
    """test_kd.cpp:99-117: rays between two uniform points of the sphere (center, 0.2), mint 0"""
    rng = np.random.default_rng(seed)
    c = np.array([-0.016840, 0.110154, -0.001537], np.float32)

    def on_sphere(u):
        z = 1.0 - 2.0 * u[:, 1]; r = np.sqrt(np.maximum(0.0, 1.0 - z * z)); phi = 2.0 * np.pi * u[:, 0]
        return np.stack([r * np.cos(phi), r * np.sin(phi), z], 1)
    p1 = c + 0.2 * on_sphere(rng.random((n, 2))); p2 = c + 0.2 * on_sphere(rng.random((n, 2)))
    d = p2 - p1; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = p1; rays[:, 3] = 0.0; rays[:, 4:7] = d; rays[:, 7] = np.inf
    return rays


def bunny_scene(V, F, gauss, w=64, h=64):
    from mitsuba_amd import scene as S
    sb = S.SceneBuilder()
    sb.mesh(V, F, sb.diffuse((0.6, 0.55, 0.5)))
    sb.quad((-0.3, 0.032, -0.3), (0.3, 0.032, -0.3), (0.3, 0.032, 0.3), (-0.3, 0.032, 0.3), sb.diffuse((0.4, 0.4, 0.45)), facing=(0, 1, 0))
    light = sb.diffuse((0, 0, 0))
    sb.quad((-0.2, 0.5, -0.2), (0.2, 0.5, -0.2), (0.2, 0.5, 0.2), (-0.2, 0.5, 0.2), light, facing=(0, -1, 0), radiance=(18.0, 17.0, 15.0))
    sb.perspective((-0.05, 0.18, 0.32), (-0.017, 0.10, 0.0), (0, 1, 0), 40.0)
    sb.hdrfilm(w, h, gauss)
    return sb


if __name__ == "__main__":
    from oracle import ref_ffi as R, oracle_ffi as O
    V, F = read_ply(os.path.join(REF, "data", "tests", "bunny.ply"))
    gauss = O.gaussian_filter(0.5)
    desc = bunny_scene(V, F, gauss).desc()
    rays = chords(40000)
    rs = R.RefScene(desc)
    hits = rs.trace(rays)                                  # (t, u, v, shape, prim) by the reference's kd-tree
    t = hits[:, 0].astype(np.float32); shape = hits[:, 3].astype(np.int32); prim = hits[:, 4].astype(np.int32)
    print("bunny: %d vertices, %d faces; %d chords, %.2f%% hit something" % (len(V), len(F), len(rays), 100 * np.isfinite(t).mean()))
    out = os.path.join(ROOT, "tests", "golden", "bunny.npz")
    np.savez_compressed(out, V=V, F=F.astype(np.int32), chord_seed=1234, n_chords=len(rays), ref_t=t, ref_uv=hits[:, 1:3].astype(np.float32),
                        ref_shape=shape.astype(np.int8), ref_prim=prim)
    print("wrote", out, os.path.getsize(out), "bytes")
