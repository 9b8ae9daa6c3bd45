#!/bin/bash
# round 3, GPU call L: k_shade with LDS-addressed tables vs flat loads
run() { # label scene spp env...
  label=$1; sc=$2; spp=$3; shift 3
  env "$@" SPP=$spp python tools/gpu_scenes.py $sc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-10s %-8s %4d spp %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f fused %6.1f wall %7.1f  iters %d' % ('$label', d['scene'], d['spp'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], k['fused_kernel_ms'], d['wall_ms'], d['iters']))"
}
run lds atrium 64 X=1
run flat atrium 64 PHIP_SHADE_FLAT_TABLES=1
run lds glass 128 X=1
run flat glass 128 PHIP_SHADE_FLAT_TABLES=1
run lds cornell 256 REPEAT=2
run wavefront cornell 256 PHIP_MEGA=0
run wf_flat cornell 256 PHIP_MEGA=0 PHIP_SHADE_FLAT_TABLES=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -x -q 2>&1 | tail -3
