"""First GPU bring-up script: ray-cast parity + Cornell render parity vs the oracle."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
from oracle import oracle_ffi as O

L = _ffi.lib()
print("devices:", L.phip_device_count(), L.phip_version())
ft = _ffi.gaussian_filter(0.5)
fo = O.gaussian_filter(0.5)
print("filter tables equal:", ft == fo)
W = int(os.environ.get("W", 128)); SPP = int(os.environ.get("SPP", 16)); MD = int(os.environ.get("MD", 4))
sb = S.cornell_box(W, W, ft)
d = sb.desc()
osc = O.OracleScene(d)
gsc = Scene(d)
print("accel:", gsc.accel_info().as_dict())

# ray casts
rng = np.random.default_rng(1)
n = 200000
o = rng.uniform(0, 550, (n, 3)).astype(np.float32)
dd = rng.normal(size=(n, 3)).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32); rays[:, :3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = dd; rays[:, 7] = np.inf
t = time.time(); gh, go, gst = gsc.rayIntersect(rays, True, True); tg = time.time() - t
t = time.time(); oh, oo, ost = osc.trace(rays, True, True); to = time.time() - t
print("raycast: gpu %.3fs oracle %.3fs" % (tg, to), "kernel ms", gst.trace_kernel_ms)
same = (gh.view(np.uint32) == oh.view(np.uint32)).all(axis=1)
print("closest bit-identical: %d / %d" % (same.sum(), n), " shadow identical:", (go == oo).sum())
bad = np.where(~same)[0][:5]
for b in bad: print("  mismatch", b, gh[b], O.hits_prim(gh)[b], oh[b], O.hits_prim(oh)[b])

# render
integ = PathHIP(maxDepth=MD)
film = HDRFilm(W, W)
t = time.time()
ok = integ.render(gsc, film, SPP, flags=A.PHIP_FLAG_SAMPLE_BUFFER | A.PHIP_FLAG_KERNEL_TIMING)
tg = time.time() - t
print("gpu render ok", ok, "%.3fs" % tg, integ.stats.as_dict())
gs = integ.samples(gsc, SPP)
p = A.default_render_params(spp=SPP, max_depth=MD)
t = time.time(); ofilm, osmp, ost = osc.render(p, want_samples=True); to = time.time() - t
print("oracle render %.3fs" % to, ost.as_dict())
eq = (gs.view(np.uint32) == osmp.view(np.uint32)).all(axis=-1)
print("samples bit-identical: %d / %d (%.6f%%)" % (eq.sum(), eq.size, 100.0 * eq.mean()))
g_rgb = film.develop(); o_rgb = O.develop(ofilm)
rel = np.linalg.norm(g_rgb - o_rgb) / np.linalg.norm(o_rgb)
print("rel L2 (developed):", rel, " max abs:", np.abs(g_rgb - o_rgb).max(), "film raw rel:", np.linalg.norm(film.storage - ofilm) / np.linalg.norm(ofilm))
if not eq.all():
    ys, xs, ks = np.where(~eq)
    for i in range(min(5, len(ys))):
        print("  diff at", ys[i], xs[i], ks[i], gs[ys[i], xs[i], ks[i]], osmp[ys[i], xs[i], ks[i]])
