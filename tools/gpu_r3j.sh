#!/bin/bash
# round 3, GPU call J: where k_mega spends its wave time (C2): phases of the loop and their lane utilisation
PHIP_LIB=$PWD/mitsuba_amd/_build/libphip_megaprof.so python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
ft = _ffi.gaussian_filter()
sb = S.cornell_box(1024, 1024, ft); sc = Scene(sb.desc()); integ = PathHIP(maxDepth=-1); film = HDRFilm(1024, 1024)
integ.render(sc, film, 1)
integ.render(sc, film, 256, flags=A.PHIP_FLAG_KERNEL_TIMING)
st = integ.stats.as_dict()
T = [st['closest_rays'] * 256, st['closest_node_visits'] * 256, st['closest_triangle_tests'] * 256, st['shadow_rays'] * 256]
Ln = [None, st['shadow_node_visits'], st['shadow_triangle_tests'], st['path_vertices']]
iters = st['samples']
tot = sum(T)
print("k_mega %.1f ms; wave iterations %d; wave ticks by phase (regeneration, closest hit, vertex, shadow ray): %s" % (st['fused_kernel_ms'], iters, ["%.3f" % (t / tot) for t in T]))
print("ticks per iteration: %s" % ["%.0f" % (t / iters) for t in T])
print("active lanes per executed phase (of 64): closest %.1f, vertex %.1f, shadow %.1f" % (Ln[1] / iters, Ln[2] / iters, Ln[3] / iters))
PY
