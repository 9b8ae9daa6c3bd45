"""debug: PHIP_SAMPLER_SOBOL / _STRATIFIED on the device vs the oracle, by path depth"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mitsuba_amd import _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
from oracle import oracle_ffi as O
from conftest import sobol_tables
O.build(); gauss = O.gaussian_filter(0.5)
for w, h in ((16, 16), (64, 64)):
    desc = S.cornell_box(w, h, gauss).desc()
    gs = Scene(desc); osc = O.OracleScene(desc)
    for name, kw in (("sobol", dict(sobol=sobol_tables(w, h))), ("strat", dict(sampler=A.PHIP_SAMPLER_STRATIFIED)), ("ctr", {})):
        for md in (1, 2, 3, 4, 8):
            for rr in (2, 5):
                integ = PathHIP(maxDepth=md, rrDepth=rr)
                film = HDRFilm(w, h)
                integ.render(gs, film, 4, flags=A.PHIP_FLAG_SAMPLE_BUFFER, **kw)
                g = integ.samples(gs, 4)
                p = integ.params(gs, 4, **kw)
                of, o, _ = osc.render(p, want_samples=True)
                same = (g.view(np.uint32) == o.view(np.uint32)).all(-1)
                print(w, name, "maxDepth", md, "rrDepth", rr, "identical %.4f" % same.mean(), "per sample index", same.mean(axis=(0, 1)).round(3), "film rel", float(np.abs(film.storage - of).max()))
