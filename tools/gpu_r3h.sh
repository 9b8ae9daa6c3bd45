#!/bin/bash
# round 3, GPU call H: k_shade state exchange through LDS + 5 waves, pool policy; then the parity suite on the new defaults
out=gpurun_out/r3h; mkdir -p $out
b=$PWD/mitsuba_amd/_build
run() { # label scene spp env...
  label=$1; sc=$2; spp=$3; shift 3
  env "$@" SPP=$spp python tools/gpu_scenes.py $sc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-10s %-8s %4d spp %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f  wall %7.1f  iters %d' % ('$label', d['scene'], d['spp'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], d['wall_ms'], d['iters']))"
}
run base atrium 64 X=1
run nosort atrium 64 PHIP_SHADE_SORT=0
run base glass 512 X=1
run base atrium4k 64 X=1
run base atrium 64 PHIP_DEBUG_TIMING=1
PHIP_DEBUG_TIMING=1 SPP=64 python tools/gpu_scenes.py atrium 2>&1 | grep "\[phip\]"
echo "== parity"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
