#!/usr/bin/env python3
"""Quality of the acceleration structure, measured on the CPU (no GPU needed): node steps and Wald tests per ray of the 8-wide tree
(the host twin of k_rays_w's traversal, phip_debug_host_trace_wide) for incoherent rays in one of the benchmark scenes -- bounce rays
(origin on a surface, direction uniform over the sphere) and bounded shadow-like segments between two surface points.
    python tools/bvh_quality.py [atrium|glass_room] [n_rays]        (builder experiment hooks: PHIP_BVH_* environment variables)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _abi as A, _ffi, scene as S          # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
phip = _ffi.lib()
gauss = _ffi.gaussian_filter(0.5)
desc = getattr(S, name)(64, 36, gauss).desc()
P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3)).copy()
T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3)).copy()


def trace(rays):
    r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    hits = np.zeros((len(r), 4), np.float32)
    info = A.phip_accel_info()
    stride = 512
    seq = np.zeros((len(r), stride), np.uint8)
    t = time.time()
    rc = phip.phip_debug_host_trace_wide(P.ctypes.data_as(C.POINTER(C.c_float)), len(P), T.ctypes.data_as(C.POINTER(C.c_uint32)), len(T),
                                         r.ctypes.data_as(C.POINTER(A.phip_ray)), len(r), hits.ctypes.data_as(C.POINTER(A.phip_hit)), 1, C.byref(info),
                                         seq.ctypes.data_as(C.POINTER(C.c_uint8)), stride)
    assert rc == 0, phip.phip_last_error()
    return hits, info, (seq == 1).sum(1), (seq == 2).sum(1), time.time() - t


rng = np.random.default_rng(7)
lo, hi = P.min(0), P.max(0)
# seed rays from inside the scene -> surface points
o = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, np.full((n, 1), 1e-4, np.float32), d, np.full((n, 1), np.inf, np.float32)], 1)
h, info, _, _, _ = trace(rays)
ok = np.isfinite(h[:, 0])
pts = (o + d * h[:, :1])[ok]
pts = pts - d[ok] * 1e-3                                    # back off the surface
m = len(pts)
d2 = rng.normal(size=(m, 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
bounce = np.concatenate([pts, np.full((m, 1), 1e-4, np.float32), d2, np.full((m, 1), np.inf, np.float32)], 1)
hb, info, nb, tb, sec = trace(bounce)
q = pts[rng.permutation(m)]
seg = q - pts; ln = np.linalg.norm(seg, axis=1, keepdims=True); seg = seg / ln
shadow = np.concatenate([pts, np.full((m, 1), 1e-4, np.float32), seg, (ln * (1 - 1e-3)).astype(np.float32)], 1)
hs, _, ns, ts, _ = trace(shadow)
cks = int(np.bitwise_xor.reduce(hb.view(np.uint32).ravel()))
print("%s: %d triangles, %d wide nodes, %d records, depth %d, SAH %.2f, build %.0f ms" % (name, len(T), info.n_nodes, info.n_triangle_refs, info.max_depth, info.sah_cost, info.build_ms))
print("bounce rays (closest hit): %.2f node steps, %.2f Wald tests per ray | segments (closest hit): %.2f node steps, %.2f Wald tests | cost(2.5 n + t) %.2f / %.2f | hits xor %08x"
      % (nb.mean(), tb.mean(), ns.mean(), ts.mean(), 2.5 * nb.mean() + tb.mean(), 2.5 * ns.mean() + ts.mean(), cks))
