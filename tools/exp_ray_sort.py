"""Experiment (round 3): how much would k_raycast_w gain from coherent ray ORDER on the Sponza-class scene?  Rays of bounce b of a diffuse
random walk from the camera, traced in (a) path-slot order (= what the slot-stable pool holds), (b) shuffled, (c) sorted by a Morton key of
the origin (5 bits per axis) + direction octant.  Upper bound for a device-side ray sort; prints kernel ms per ordering and bounce."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _abi as A, _ffi, scene as S
from mitsuba_amd.integrator import Scene

gauss = _ffi.gaussian_filter(0.5)
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H = 1920, 1080
sb = (S.atrium if name == "atrium" else S.glass_room)(W, H, gauss)
desc = sb.desc()
pos = np.concatenate(sb.positions).astype(np.float32)
v0 = np.cumsum([0] + [len(p) for p in sb.positions])[:-1]
idx = np.concatenate([t + np.uint32(o) for t, o in zip(sb.indices, v0)]).astype(np.int64)
gn = np.cross(pos[idx[:, 1]] - pos[idx[:, 0]], pos[idx[:, 2]] - pos[idx[:, 0]])
gn /= np.maximum(np.linalg.norm(gn, axis=1, keepdims=True), 1e-30)
gs = Scene(desc)
rng = np.random.default_rng(1)
SPP = int(os.environ.get("SPP", "2"))
# camera rays in 8x8-tile order (a wave = a tile), SPP per pixel
ty, tx = np.meshgrid(np.arange(H // 8), np.arange(W // 8), indexing="ij")
iy, ix = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
px = (tx[..., None, None] * 8 + ix).reshape(-1); py = (ty[..., None, None] * 8 + iy).reshape(-1)
px = np.repeat(px, SPP).astype(np.float32) + rng.random(len(px) * SPP, dtype=np.float32)
py = np.repeat(py, SPP).astype(np.float32) + rng.random(len(py) * SPP, dtype=np.float32)
M = np.array(list(desc.camera.to_world), np.float32).reshape(4, 4)
tanx = np.tan(np.radians(desc.camera.xfov_deg) / 2)
dc = np.stack([(0.5 - px / W) * 2 * tanx, (0.5 - py / H) * 2 * tanx * H / W, np.ones_like(px)], 1)
dc /= np.linalg.norm(dc, axis=1, keepdims=True)
d = dc @ M[:3, :3].T
o = np.broadcast_to(M[:3, 3], d.shape).copy()
n = len(d)
bmin, bmax = pos.min(0), pos.max(0)

def part1by2(x):
    x = x.astype(np.uint32) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
    return x

def sort_key(o, d, bits=5):
    q = np.clip(((o - bmin) / (bmax - bmin) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    m = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
    octant = (d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2)
    return (m.astype(np.uint64) << 3) | octant

def trace(rays, reps=3):
    best = 1e9
    for _ in range(reps):
        hits, _, st = gs.rayIntersect(rays, True, False)
        best = min(best, st.trace_kernel_ms)
    return hits, best, st

alive = np.ones(n, bool)
for bounce in range(5):
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o; rays[:, 3] = 1e-3 if bounce else 0.0; rays[:, 4:7] = d; rays[:, 7] = np.where(alive, np.inf, 0.0)   # dead slots: an empty interval (the pool keeps them too)
    hits, t_slot, st = trace(rays)
    live = np.flatnonzero(alive)
    lr = rays[live]
    _, t_live, _ = trace(lr)
    perm = rng.permutation(len(lr))
    _, t_shuf, _ = trace(lr[perm])
    out = [("slot order incl. dead slots", t_slot), ("live rays, slot order", t_live), ("live rays, shuffled", t_shuf)]
    for bits in (3, 5, 7):
        k = sort_key(lr[:, 0:3], lr[:, 4:7], bits)
        order = np.argsort(k, kind="stable")
        _, t_sorted, _ = trace(lr[order])
        out.append(("sorted, %d bits/axis + octant" % bits, t_sorted))
    print("%s bounce %d: %d live rays of %d slots; " % (name, bounce, len(live), n) + "; ".join("%s %.2f ms" % x for x in out)
          + "; nodes/ray %.1f" % (st.closest_node_visits / max(1, st.closest_rays)), flush=True)
    # next bounce: cosine-weighted direction about the geometric normal facing the incoming ray
    t = hits[:, 0]; prim = hits[:, 3].view(np.uint32)
    hit = alive & (prim != 0xFFFFFFFF) & np.isfinite(t)
    p = o + t[:, None] * d
    nn = gn[np.where(hit, prim, 0).astype(np.int64)].astype(np.float32)
    nn = np.where((np.sum(nn * d, 1) > 0)[:, None], -nn, nn)
    u1, u2 = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    r = np.sqrt(u1); phi = 2 * np.pi * u2
    a = np.where(np.abs(nn[:, 0:1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    tt = np.cross(a, nn); tt /= np.linalg.norm(tt, axis=1, keepdims=True); bb = np.cross(nn, tt)
    d = (r * np.cos(phi))[:, None] * tt + (r * np.sin(phi))[:, None] * bb + np.sqrt(np.maximum(0, 1 - u1))[:, None] * nn
    d = d.astype(np.float32); o = np.where(hit[:, None], p, o).astype(np.float32)
    alive = hit & (rng.random(n) < 0.75)                   # ~ the survival rate of the atrium's paths per vertex
