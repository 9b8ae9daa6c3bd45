#!/usr/bin/env python3
"""Stress of the fused kernel on the tree in memory (k_wide_wave.h: one shared task stack per wave in LDS, pair queue, spill buffer): the same frames rendered
again and again -- every render must stay on the fused kernel (a pass that gave up would be re-rendered by the wavefront kernels: stats.fused says so), deliver
every sample and the same bits (a task lost or run twice, a stale slot, a race on the wave's buffers would change a sample).
    python tools/fused_wide_stress.py [repeats]        (on a GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, VolPathSimpleHIP, HDRFilm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ft = _ffi.gaussian_filter()
t0 = time.time(); renders = 0
cases = (("spheres 1k, glass + copper", lambda w, h: S.cornell_spheres(w, h, ft, nlon=24, nlat=12), 0),
         ("spheres 1k, diffuse", lambda w, h: S.cornell_spheres(w, h, ft, nlon=24, nlat=12, materials=False), 0),
         ("spheres 18k", lambda w, h: S.cornell_spheres(w, h, ft, nlon=96, nlat=48), 0),
         ("82 records", lambda w, h: S.cornell_box(w, h, ft, extra_blocks=5), 0),
         ("atrium, fused by request", lambda w, h: S.atrium(w, h, ft, detail=0.3), A.PHIP_FLAG_FUSED_ANY),
         ("glass room, fused by request", lambda w, h: S.glass_room(w, h, ft, detail=0.3), A.PHIP_FLAG_FUSED_ANY))
for name, build, flags in cases:
    for w, h, spp, reps in ((512, 512, 16, n), (100, 70, 3, 4 * n), (64, 64, 1, 4 * n)):
        sc = Scene(build(w, h).desc())
        for integ in (PathHIP(maxDepth=-1), PathHIP(maxDepth=12, strictNormals=True), DirectHIP(shadingSamples=2), VolPathSimpleHIP(maxDepth=8)):
            ref = None
            for i in range(reps):
                film = HDRFilm(w, h)
                assert integ.render(sc, film, spp, flags=flags)
                assert integ.stats.fused == 1 and integ.stats.samples == w * h * spp, (name, integ.stats.as_dict())
                if ref is None: ref = film.storage.copy()
                else: assert (film.storage.view(np.uint32) == ref.view(np.uint32)).all(), (name, w, h, spp, i)
                renders += 1
        sc.close()
    print("%-32s ok (%d renders so far, %.1f s)" % (name, renders, time.time() - t0), flush=True)
print("%d renders on the fused kernel, every sample delivered, every frame bit-identical to its first render, %.1f s" % (renders, time.time() - t0))
