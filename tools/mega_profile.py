#!/usr/bin/env python3
"""Phase profile of k_mega's persistent loop from a -DMEGA_PROFILE=1 build (tools/build_variant.sh prof with MEGA_FLAGS):
the work-counter rows of phip_stats carry wave-clock ticks and active-lane counts per phase instead (k_mega.h, end of the kernel).

    PHIP_LIB=mitsuba_amd/_build/libphip_prof.so SPP=64 [SCENE=cornell_mixed] python tools/mega_profile.py [out.json]
(the exchange of MEGA_CLASS_DEAL lies between the closest-hit and the vertex phase and is in neither: it is what the four shares leave of the kernel's time)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _ffi, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm

spp = int(os.environ.get("SPP", 64))
w = h = 1024
scene = os.environ.get("SCENE", "cornell_box")
sc = Scene(getattr(S, scene)(w, h, _ffi.gaussian_filter(), **json.loads(os.environ.get("SCENE_KW", "{}"))).desc())      # e.g. SCENE=cornell_spheres SCENE_KW='{"materials": false}'
integ = PathHIP(maxDepth=-1); film = HDRFilm(w, h)
integ.render(sc, film, 1)
integ.render(sc, film, spp)
st = integ.stats.as_dict()
ticks = [st["closest_rays"], st["closest_node_visits"], st["closest_triangle_tests"], st["shadow_rays"]]     # >> 8 each
lanes = [None, st["shadow_node_visits"], st["shadow_triangle_tests"], st["path_vertices"]]
iters = st["samples"]
tot = float(sum(ticks))
names = ["regeneration", "closest hit", "vertex", "shadow ray"]
out = {"scene": scene, "spp": spp, "wave_iterations": iters, "fused_kernel_ms": st["fused_kernel_ms"], "lib": os.environ.get("PHIP_LIB", "")}
for i, n in enumerate(names):
    out[n] = {"share": round(ticks[i] / tot, 4), "ticks_per_iteration": round(ticks[i] * 256.0 / max(iters, 1), 1),
              "lanes_per_iteration": round(lanes[i] / max(iters, 1), 2) if lanes[i] is not None else None}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
