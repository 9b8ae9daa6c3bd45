#!/bin/bash
# round 5, GPU call b: k_shade_trace (vertex + shadow ray + next ray of a slot in one kernel per iteration, small scenes that are not k_mega's) against the
# three-kernel iterations (PHIP_NO_SHADE_TRACE=1) on the mixed Cornell box; parity of the scenes that now take it   -> gpurun_out/r5b/
mkdir -p gpurun_out/r5b
o=gpurun_out/r5b
rm -f mitsuba_amd/_build/libphip_*.so
WORKLOADS="cornell 256;cmixed 256" AB_ENV="classic PHIP_NO_SHADE_TRACE=1" bash tools/gpu_ab.sh > $o/ab_small.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $o/pytest_parity.txt 2>&1
tail -15 $o/pytest_parity.txt; cat $o/ab_small.txt
