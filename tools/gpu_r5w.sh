#!/bin/bash
# round 5, GPU call w: the mid-sized scenes -- the Cornell box with a glass and a copper sphere (or two diffuse spheres) of 1 k / 4.5 k / 18 k
# triangles, 1024 x 1024 x 64 spp, maxDepth -1: which device path renders them, at what rate, with what kernel split   -> gpurun_out/r5w/
mkdir -p gpurun_out/r5w
for s in cornell cmixed sph1kd sph18kd sph1k sph5k sph18k; do
  REPEAT=2 SPP=64 python tools/gpu_scenes.py $s 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']; a=d['accel']
print('%-8s %6d tris  fused %d  fits_lds %s wide-nodes %s  %7.1f Msamples/s  %6.1f Mrays/s  len %.2f  rays %6.1f ms  shade %6.1f ms  film %4.1f  fused %6.1f  wall %6.1f  iters %3d  nodes/closest %.1f tris/closest %.1f nodes/shadow %.1f' % (d['scene'], d['tris'], d['fused'], a.get('fits_lds'), a.get('wide_nodes', a.get('n_wide_nodes')), d['Msamples/s'], d['Mrays/s'], d['mean_len'], k['trace_kernel_ms']+k['shadow_kernel_ms'], k['shade_kernel_ms'], k['film_kernel_ms'], k['fused_kernel_ms'], d['wall_ms'], d['iters'], d['nodes/closest'], d['tris/closest'], d['nodes/shadow']))"
done 2>&1 | tee gpurun_out/r5w/mid_sized_scenes.txt
