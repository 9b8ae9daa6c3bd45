#!/bin/bash
# round 3, GPU call I: k_film_tiled2 vs the round-2 tiled film kernel: bit identity of the film and time, C2; film-related parity tests
python - <<'PY'
import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
ft = _ffi.gaussian_filter()
out = {}
for name, w, h, spp, md in (("cornell_box", 1024, 1024, 256, -1), ("cornell_box", 333, 217, 8, 5), ("atrium", 1920, 1080, 16, 8)):
    sb = getattr(S, name)(w, h, ft); sc = Scene(sb.desc()); integ = PathHIP(maxDepth=md); film = HDRFilm(w, h)
    integ.render(sc, film, 1)
    integ.render(sc, film, spp, flags=A.PHIP_FLAG_KERNEL_TIMING)
    st = integ.stats.as_dict()
    np.save(os.environ["OUT"] + "_%s_%d.npy" % (name, w), film.storage)
    print(name, w, h, spp, "film %.3f ms fused %.1f ms render %.1f ms" % (st["film_kernel_ms"], st["fused_kernel_ms"], st["render_ms"]))
'''
for tag, env in (("v2", {}), ("v1", {"PHIP_FILM_V1": "1"})):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OUT="/tmp/film_" + tag, **env), capture_output=True, text=True)
    print(tag, r.stdout, r.stderr[-500:])
for f in ("cornell_box_1024", "cornell_box_333", "atrium_1920"):
    a, b = np.load("/tmp/film_v1_%s.npy" % f), np.load("/tmp/film_v2_%s.npy" % f)
    print(f, "bit-identical film:", bool((a.view(np.uint32) == b.view(np.uint32)).all()), "max abs diff", float(np.abs(a - b).max()))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cornell or ragged or block_sizes or shards or filter_widths or empty" 2>&1 | tail -3
