#!/bin/bash
# round 3, GPU call D: rays clipped by the shading kernels (preclip) -- A/B against the build before it, refill thresholds, 5 waves, refill profile, parity
out=gpurun_out/r3d; mkdir -p $out
b=$PWD/mitsuba_amd/_build
run() { # label env...
  label=$1; shift
  for s in "atrium 64" "glass 128"; do set -- $s "$@"
    env "${@:3}" SPP=$2 python tools/gpu_scenes.py $1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-14s %-7s %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f  wall %7.1f  iters %d' % ('$label', d['scene'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], d['wall_ms'], d['iters']))"
    shift 2; done; }
./tools/lds_dma_test
run before PHIP_LIB=$b/libphip_before.so
run preclip X=1
run r8 PHIP_LIB=$b/libphip_r8.so
run r12 PHIP_LIB=$b/libphip_r12.so
run w5 PHIP_LIB=$b/libphip_w5.so
run preclip2 X=1
echo "== refill profile"
PHIP_LIB=$b/libphip_prof.so python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
ft = _ffi.gaussian_filter()
sb = S.atrium(1920, 1080, ft); sc = Scene(sb.desc()); integ = PathHIP(maxDepth=8); film = HDRFilm(1920, 1080)
integ.render(sc, film, 64, flags=A.PHIP_FLAG_KERNEL_TIMING)
st = integ.stats.as_dict()
tot, refill, refills, iters, assign, load = st['shadow_triangle_tests'], st['shadow_node_visits'], st['shadow_rays'], st['closest_triangle_tests'], st['closest_node_visits'], st['closest_rays']
print("PROFILE atrium: total wave ticks %d; refill %.3f of it (assign %.3f, load wait (slowest lane) %.3f, rest %.3f); %d refills, %.2f iterations per refill; ticks per iteration %.0f, per refill %.0f (assign %.0f, load %.0f)" % (
    tot, refill / tot, assign / tot, load / tot, (refill - assign - load) / tot, refills, iters / refills, (tot - refill) / iters, refill / refills, assign / refills, load / refills))
PY
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_round2.py tests/test_bunny.py -x -q 2>&1 | tail -5
