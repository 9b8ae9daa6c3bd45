#!/usr/bin/env python3
"""Stress of k_mega's mailbox protocol (k_mega.h: MEGA_MAILBOX): the same frames rendered again and again -- every render must deliver every sample (the
statistics count them; a wave that gives up waiting makes phip_render fail) and the same bits (the film's sums are ordered: a path that went missing or
was shaded twice would show).    python tools/mailbox_stress.py [repeats]        (on a GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mitsuba_amd import _ffi, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, VolPathSimpleHIP, HDRFilm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# SAMPLER=sobol|halton: the QMC builds of the mailbox kernel (round 6: the sample's sequence index travels with the path through the mailboxes)
skw = lambda w, h: {}
if os.environ.get("SAMPLER") in ("sobol", "halton"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import conftest as CT
    from mitsuba_amd import _abi as A
    skw = (lambda w, h: dict(sobol=CT.sobol_tables(w, h))) if os.environ["SAMPLER"] == "sobol" else (lambda w, h: dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=CT.qmc_tables(-1)))
ft = _ffi.gaussian_filter()
t0 = time.time(); renders = 0
for w, h, spp, reps, kw in ((1024, 1024, 32, n, {}), (256, 256, 8, 4 * n, {}), (100, 70, 3, 8 * n, {}), (64, 64, 1, 8 * n, {}), (512, 512, 16, n, dict(strictNormals=True, maxDepth=12))):
    for build in (S.cornell_mixed, lambda w, h, f: S.cornell_box(w, h, f, short_bsdf=lambda b: b.twosided(b.roughconductor(S.CU_ETA, S.CU_K, alpha=0.1)))):
        sc = Scene(build(w, h, ft).desc())
        for cls in (PathHIP, VolPathSimpleHIP):
            integ = cls(**({"maxDepth": -1} | kw)); ref = None
            for i in range(reps):
                film = HDRFilm(w, h)
                assert integ.render(sc, film, spp, **skw(w, h))
                assert integ.stats.fused == 1 and integ.stats.samples == w * h * spp, integ.stats.as_dict()
                if ref is None: ref = film.storage.copy()
                else: assert (film.storage.view(np.uint32) == ref.view(np.uint32)).all(), (w, h, spp, i)
                renders += 1
        sc.close()
print("%d renders, every sample delivered, every frame bit-identical to its first render, %.1f s" % (renders, time.time() - t0))
