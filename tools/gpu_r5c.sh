#!/bin/bash
# round 5, GPU call c: k_shade_trace with its lanes dealt by BSDF model (PHIP_SHADE_SORT=0: without) on the mixed Cornell box; parity   -> gpurun_out/r5c/
mkdir -p gpurun_out/r5c
o=gpurun_out/r5c
rm -f mitsuba_amd/_build/libphip_*.so
WORKLOADS="cmixed 256" AB_ENV="nosort PHIP_SHADE_SORT=0;classic PHIP_NO_SHADE_TRACE=1" bash tools/gpu_ab.sh > $o/ab_small.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "mixed or zoo or record or cornell_render or constant_env or textures or envmap" > $o/pytest_parity.txt 2>&1
tail -5 $o/pytest_parity.txt; cat $o/ab_small.txt
