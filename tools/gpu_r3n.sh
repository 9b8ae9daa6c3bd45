#!/bin/bash
# round 3, GPU call N: k_shade at 5 / 6 waves on the glass room and the atrium; the atrium's kd-tree residue sample by sample; the CLI drop-in test
b=$PWD/mitsuba_amd/_build
run() { # label scene spp env...
  label=$1; sc=$2; spp=$3; shift 3
  env "$@" SPP=$spp python tools/gpu_scenes.py $sc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-10s %-8s %4d spp %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f fused %6.1f wall %7.1f  iters %d' % ('$label', d['scene'], d['spp'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], k['fused_kernel_ms'], d['wall_ms'], d['iters']))"
}
run base glass 128 X=1
run s5 glass 128 PHIP_LIB=$b/libphip_s5.so
run s6 glass 128 PHIP_LIB=$b/libphip_s6.so
run base atrium 64 X=1
run s5 atrium 64 PHIP_LIB=$b/libphip_s5.so
run s6 atrium 64 PHIP_LIB=$b/libphip_s6.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "atrium" 2>&1 | grep -v "^$" | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "crop" 2>&1 | grep -v "^$" | tail -6
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -s -k "scene_file" 2>&1 | grep -v "^$" | tail -12
