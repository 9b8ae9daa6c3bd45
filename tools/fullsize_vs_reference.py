#!/usr/bin/env python3
"""BASELINE configs C2 / C3 / C4 at full size: path_hip on the GPU against Mitsuba 0.6 itself (oracle/_ref, all host cores,
parity-stream sampler) -- developed images, relative L2.
    python tools/fullsize_vs_reference.py out.json [C2|C2sobol|C3|C4-class|C4res|C4full]          (on a GPU box)
    LD_PRELOAD=$PWD/oracle/_build/libcrm.so python tools/fullsize_vs_reference.py ...  the same against the reference with the
        correctly rounded transcendentals of include/phip_fmath.h in place of glibc's (oracle/ref_glue/crlibm_shim.cpp)"""
import json
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _abi as A, _ffi, scene as S          # noqa: E402
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm    # noqa: E402
from oracle import ref_ffi as R                               # noqa: E402

gauss = _ffi.gaussian_filter(0.5)
out = {}
# which libm answers the reference's transcendentals in this process (tests/test_gpu_dropin.py puts oracle/_ref/pinned_libm first on LD_LIBRARY_PATH)
LIBM_MAPPED = next((line.split()[-1] for line in open("/proc/self/maps") if "/libm.so" in line or "/libm-" in line), "?")
_v = ctypes.CDLL(None).gnu_get_libc_version; _v.restype = ctypes.c_char_p
GLIBC_VERSION = _v().decode()
from oracle import oracle_ffi as O                             # noqa: E402
for name, build, w, h, spp, md in (("C2 cornell 1024x1024x256", S.cornell_box, 1024, 1024, 256, -1),
                                   ("C2sobol cornell 1024x1024x256, the reference's own sobol sampler on both sides", S.cornell_box, 1024, 1024, 256, -1),
                                   ("C3 atrium 1920x1080x64", S.atrium, 1920, 1080, 64, 8),
                                   ("C4-class glass room 960x540x64 md16", S.glass_room, 960, 540, 64, 16),
                                   ("C4res glass room 1920x1080x64 md16", S.glass_room, 1920, 1080, 64, 16),
                                   ("C4full glass room 1920x1080x512 md16", S.glass_room, 1920, 1080, 512, 16)):
    if len(sys.argv) > 2 and not any(k in name for k in sys.argv[2:]):
        continue
    if "C2sobol" in name and "C2sobol" not in sys.argv[2:]:
        continue                                              # (no glue sampler at all: <sampler type="sobol"/> is deterministic; only on request)
    if "C4res" in name and "C4res" not in sys.argv[2:]:
        continue                                              # (C4's frame and depth at 1/8 of its samples per pixel: the driver-run suite asks for it)
    if "C4full" in name and "C4full" not in sys.argv[2:]:
        continue                                              # ~10 minutes of the reference on 256 threads: only on request
    desc = build(w, h, gauss).desc()
    gs = Scene(desc)
    film = HDRFilm(w, h)
    integ = PathHIP(maxDepth=md)
    kw = dict(sobol=R.sobol_tables(w, h)) if "C2sobol" in name else {}      # the plugin's direction numbers, read out of the loaded sobol.so
    integ.render(gs, film, spp, **kw)                         # warm-up
    film = HDRFilm(w, h)
    t = time.time(); assert integ.render(gs, film, spp, **kw); tg = time.time() - t
    g = film.develop()
    rs = R.RefScene(desc)
    cpu, sec = rs.render_job(A.default_render_params(spp=spp, max_depth=md), sampler="sobol" if "C2sobol" in name else "ctr")
    rel = float(np.linalg.norm(g - cpu) / np.linalg.norm(cpu))
    big = float((np.abs(g - cpu) > 1e-3 * np.maximum(1.0, np.abs(cpu))).any(-1).mean())
    out[name] = {"gpu_seconds": round(tg, 3), "reference_seconds": round(sec, 1), "reference_threads": os.cpu_count(), "rel_l2": rel,
                 "pixels_differing_by_more_than_1e-3": big, "speedup": round(sec / tg, 1),
                 "reference_libm": "phip_fmath.h (LD_PRELOAD libcrm.so)" if "libcrm" in os.environ.get("LD_PRELOAD", "") else "glibc",
                 "libm_mapped": LIBM_MAPPED, "glibc_version": GLIBC_VERSION}
    print(name, out[name], flush=True)
    rs.close(); gs.close()
os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
