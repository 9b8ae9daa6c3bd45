#!/bin/bash
# A/B of library variants on the GPU box, one gpurun call: the product first, then every mitsuba_amd/_build/libphip_*.so built by
# tools/build_variant.sh (same sources, extra -D flags), each on the workloads below.  Extra environment for a row: A/B rows of the form
# "label VAR=value ..." in the AB_ENV variable, separated by ';' (e.g. AB_ENV="nosort PHIP_SHADE_SORT=0;pool8M PHIP_POOL=8388608").
#   WORKLOADS="atrium 64;glass 128;cornell 256" bash tools/gpu_ab.sh
# (round 3's one-off experiment scripts gpu_r3a..q.sh were folded into this one; their outputs are profiles/r03_gpu_call_logs.txt)
b=$PWD/mitsuba_amd/_build
WORKLOADS=${WORKLOADS:-"atrium 64;glass 128"}
run() { # label env...
  label=$1; shift
  IFS=';' read -ra W <<< "$WORKLOADS"
  for s in "${W[@]}"; do set -- $s "$@"
    env "${@:3}" SPP=$2 python tools/gpu_scenes.py $1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-14s %-8s %4d spp %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f  fused %6.1f  wall %7.1f  iters %d  nodes/closest %.1f tris/closest %.1f' % ('$label', d['scene'], d['spp'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], k['fused_kernel_ms'], d['wall_ms'], d['iters'], d['nodes/closest'], d['tris/closest']))"
    shift 2; done; }
run product X=1
for l in $b/libphip_*.so; do [ -e "$l" ] && run $(basename $l .so | sed 's/libphip_//') PHIP_LIB=$l; done
IFS=';' read -ra ROWS <<< "$AB_ENV"
for r in "${ROWS[@]}"; do [ -n "$r" ] && run $r; done
run product2 X=1
