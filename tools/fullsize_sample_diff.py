#!/usr/bin/env python3
"""C2 at full size, sample by sample: the GPU against the oracle (ray queries by a sweep over every triangle, or the reference's
kd-tree with KD=1) in slices of the sample index through sample_offset / sample_total, so that the stream keys are those of the
full job.  Prints every sample whose radiance differs.   python tools/fullsize_sample_diff.py out.json [spp_per_slice] [slices]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _abi as A, _ffi, scene as S          # noqa: E402
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm    # noqa: E402
from oracle import oracle_ffi as O                             # noqa: E402

gauss = _ffi.gaussian_filter(0.5)
W = 1024; TOTAL = 256
step = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nsl = int(sys.argv[3]) if len(sys.argv) > 3 else TOTAL // step
desc = S.cornell_box(W, W, gauss).desc()
gs = Scene(desc); integ = PathHIP()
osc = O.OracleScene(desc); osc.set_bruteforce(not os.environ.get("KD"))
found = []
for sl in range(nsl):
    film = HDRFilm(W, W)
    flags = A.PHIP_FLAG_SAMPLE_BUFFER | (A.PHIP_FLAG_NO_FUSED if os.environ.get("NO_FUSED") else 0)
    assert integ.render(gs, film, step, flags=flags, sample_offset=sl * step, sample_total=TOTAL)
    gsmp = integ.samples(gs, step)
    p = integ.params(gs, step, sample_offset=sl * step, sample_total=TOTAL)
    _, osmp, _ = osc.render(p, want_samples=True)
    diff = np.argwhere((gsmp.view(np.uint32) != osmp.view(np.uint32)).any(-1))
    for y, x, j in diff:
        rec = {"x": int(x), "y": int(y), "k": int(sl * step + j), "gpu": [float(v) for v in gsmp[y, x, j]], "oracle": [float(v) for v in osmp[y, x, j]]}
        found.append(rec); print(json.dumps(rec), flush=True)
    print("slice %d/%d: %d differing samples so far" % (sl + 1, nsl, len(found)), flush=True)
json.dump(found, open(sys.argv[1], "w"), indent=1)
