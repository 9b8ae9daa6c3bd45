#!/usr/bin/env python3
"""Run under  LD_PRELOAD=oracle/_build/libcrm.so : the REFERENCE's own `path` integrator (oracle/_ref), with its libm calls answered
by the correctly rounded functions of include/phip_fmath.h, against the oracle's parity build (= the arithmetic of the GPU
kernels, bit for bit) on the same counter-based sample stream, sample by sample.  Prints one JSON line per scene.

    LD_PRELOAD=$PWD/oracle/_build/libcrm.so python tools/ref_with_cr_libm.py [scene ...]

What it shows: with the transcendentals equal, Mitsuba 0.6 and path_hip produce the same bits -- the libm rounding differences
are the ONLY source of the image differences reported for the stock reference."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mitsuba_amd import _abi as A, scene as S          # noqa: E402
from oracle import oracle_ffi as O, ref_ffi as R       # noqa: E402

gauss = O.gaussian_filter(0.5)
SCENES = {
    "cornell": (lambda: S.cornell_box(96, 96, gauss).desc(), 32, -1),
    "atrium": (lambda: S.atrium(160, 90, gauss, detail=0.5).desc(), 16, 8),
    "glass_room": (lambda: S.glass_room(120, 68, gauss, detail=0.5).desc(), 16, 16),
}
for name in (sys.argv[1:] or list(SCENES)):
    build, spp, md = SCENES[name]
    desc = build()
    p = A.default_render_params(spp=spp, max_depth=md, block_size=256)
    osc = O.OracleScene(desc)                               # parity build: phip_fmath.h
    ofilm, osmp, _ = osc.render(p, want_samples=True)
    rs = R.RefScene(desc)
    rfilm, rsmp = rs.render(p, sampler="ctr")
    same = (osmp.view(np.uint32) == rsmp.view(np.uint32)).all(-1)
    rel = float(np.linalg.norm(ofilm.astype(np.float64) - rfilm) / np.linalg.norm(rfilm))
    print(json.dumps({"scene": name, "spp": spp, "samples": int(same.size), "bit_identical": float(same.mean()),
                      "differing_samples": int((~same).sum()), "film_rel_l2": rel, "preload": os.environ.get("LD_PRELOAD", "")}), flush=True)
    rs.close(); osc.close()
