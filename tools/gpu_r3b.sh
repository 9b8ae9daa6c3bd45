#!/bin/bash
# round 3, GPU call B: what is the vector-memory instruction worth?  dummy loads / 128-byte nodes / non-temporal pool accesses; TCP counters of the shipping build
out=gpurun_out/r3b; mkdir -p $out
b=$PWD/mitsuba_amd/_build
run() { # label env...
  label=$1; shift
  for s in "atrium 64" "glass 128"; do set -- $s "$@"
    env "${@:3}" SPP=$2 python tools/gpu_scenes.py $1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-14s %-7s %7.1f Msamples/s  rays %7.1f ms  shade %6.1f ms  film %5.1f  wall %7.1f  iters %d  nodes/closest %.1f tris/closest %.1f nodes/shadow %.1f build %.2fs' % ('$label', d['scene'], d['Msamples/s'], k['trace_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], d['wall_ms'], d['iters'], d['nodes/closest'], d['tris/closest'], d['nodes/shadow'], d['scene_create_s']))"
    shift 2; done; }
echo "== vmem roof (extra modes)"; timeout 300 python tools/vmem_roof.py $out/vmem_roof.json 2>&1 | grep "of_64\|lane_random_16B \|lane_random_4B\|80B"
echo "== A/B"
run base X=1
run eager PHIP_LIB=$b/libphip_eager.so
run dummy1 PHIP_LIB=$b/libphip_dummy1.so
run dummy2 PHIP_LIB=$b/libphip_dummy2.so
run stride8 PHIP_LIB=$b/libphip_stride8.so
run nt PHIP_LIB=$b/libphip_nt.so
run eager2 PHIP_LIB=$b/libphip_eager.so
echo "== TCP counters, atrium 16 spp, shipping build"
SPP=16 bash tools/pmc_mem.sh atrium $out/pmc atrium
python tools/pmc_tcp.py $out/pmc atrium $out/tcp_atrium.json $out/vmem_roof.json
rm -f $out/pmc/*kernel_trace.csv
