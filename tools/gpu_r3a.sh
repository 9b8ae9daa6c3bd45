#!/bin/bash
# round 3, GPU call A: vmem roof micro-benchmark; refill-time profile of k_rays_w; A/B of k_shade's material deal, eager ray loads,
# refill thresholds and the tree re-optimiser on C3 / C4; parity of the new hit-class path
out=gpurun_out/r3a; mkdir -p $out
b=$PWD/mitsuba_amd/_build
run() { # label env...
  label=$1; shift
  for s in "atrium 64" "glass 128"; do set -- $s "$@"
    env "${@:3}" SPP=$2 python tools/gpu_scenes.py $1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-14s %-7s %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f  wall %7.1f  iters %d  nodes/closest %.1f tris/closest %.1f nodes/shadow %.1f build %.2fs' % ('$label', d['scene'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], d['wall_ms'], d['iters'], d['nodes/closest'], d['tris/closest'], d['nodes/shadow'], d['scene_create_s']))"
    shift 2; done; }
echo "== vmem roof"; timeout 300 python tools/vmem_roof.py $out/vmem_roof.json 2>&1 | tail -70
echo "== A/B"
run base X=1
run nosort PHIP_SHADE_SORT=0
run bvhopt0 PHIP_BVH_OPT=0
run eager PHIP_LIB=$b/libphip_eager.so
run r24 PHIP_LIB=$b/libphip_r24.so
run r32 PHIP_LIB=$b/libphip_r32.so
run base2 X=1
echo "== refill profile (rows: shadow_node_visits = refill ticks, shadow_triangle_tests = total ticks, shadow_rays = refills, closest_triangle_tests = wave iterations)"
PHIP_LIB=$b/libphip_prof.so SPP=64 python tools/gpu_scenes.py atrium 2>&1 | tail -1 > $out/prof_atrium.json; python - <<PY
import json
d=json.load(open('$out/prof_atrium.json')); print(d)
PY
PHIP_LIB=$b/libphip_prof.so SPP=64 python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
ft = _ffi.gaussian_filter()
sb = S.atrium(1920, 1080, ft); sc = Scene(sb.desc()); integ = PathHIP(maxDepth=8); film = HDRFilm(1920, 1080)
integ.render(sc, film, 64, flags=A.PHIP_FLAG_KERNEL_TIMING)
st = integ.stats.as_dict()
print("PROFILE atrium: refill ticks %d, total ticks %d => %.3f of the wave time in the refill branch; refills %d, wave iterations %d => %.2f iterations per refill; ticks per iteration %.0f, ticks per refill %.0f; rays %d" % (
    st['shadow_node_visits'], st['shadow_triangle_tests'], st['shadow_node_visits'] / max(1, st['shadow_triangle_tests']), st['shadow_rays'], st['closest_triangle_tests'],
    st['closest_triangle_tests'] / max(1, st['shadow_rays']), (st['shadow_triangle_tests'] - st['shadow_node_visits']) / max(1, st['closest_triangle_tests']), st['shadow_node_visits'] / max(1, st['shadow_rays']), st['closest_rays']))
PY
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -x -q -k "atrium or glass_room or material_zoo or raycast or bitmap" 2>&1 | tail -5
