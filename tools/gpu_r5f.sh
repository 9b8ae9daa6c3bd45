#!/bin/bash
# round 5, GPU call f: k_shade_trace as adopted (4 M-slot pool, five waves, tables in dynamic LDS) -- static share of the sample schedule; parity  -> gpurun_out/r5f/
mkdir -p gpurun_out/r5f
o=gpurun_out/r5f
rm -f mitsuba_amd/_build/libphip_*.so
WORKLOADS="cmixed 256" AB_ENV="static25 PHIP_STATIC_PERCENT=25;static50 PHIP_STATIC_PERCENT=50;static60 PHIP_STATIC_PERCENT=60;pool2M PHIP_POOL=2097152;pool8M PHIP_POOL=8388608;classic PHIP_NO_SHADE_TRACE=1" bash tools/gpu_ab.sh > $o/ab.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "mixed or zoo or record or cornell_render or constant_env or textures or envmap or fuzz" > $o/pytest_parity.txt 2>&1
tail -5 $o/pytest_parity.txt; cat $o/ab.txt
