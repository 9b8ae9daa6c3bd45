#!/bin/bash
# round 3, GPU call G: pool size sweep (per-launch drain of k_rays_w vs pass-level drain)
run() { # label scene spp env...
  label=$1; sc=$2; spp=$3; shift 3
  env "$@" SPP=$spp python tools/gpu_scenes.py $sc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-10s %-8s %4d spp %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f  wall %7.1f  iters %d' % ('$label', d['scene'], d['spp'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], d['wall_ms'], d['iters']))"
}
for p in 4194304 8388608 16777216 33554432; do run pool$((p>>20))M atrium 64 PHIP_POOL=$p; done
for p in 8388608 16777216 33554432; do run pool$((p>>20))M glass 512 PHIP_POOL=$p; done
for p in 8388608 16777216 33554432; do run pool$((p>>20))M atrium4k 64 PHIP_POOL=$p; done
for p in 4194304 8388608 16777216; do run pool$((p>>20))M atrium 16 PHIP_POOL=$p; done
