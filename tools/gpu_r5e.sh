#!/bin/bash
# round 5, GPU call e: k_shade_trace on the mixed Cornell box: pool size (its launches have no persistent ray kernel to amortise: a smaller pool shortens the
# drain of the long glass paths), five waves per SIMD (-DSHADE_TRACE_WAVES=5)   -> gpurun_out/r5e/
mkdir -p gpurun_out/r5e
o=gpurun_out/r5e
WORKLOADS="cmixed 256" AB_ENV="pool1M PHIP_POOL=1048576;pool2M PHIP_POOL=2097152;pool4M PHIP_POOL=4194304;pool8M PHIP_POOL=8388608;static50 PHIP_STATIC_PERCENT=50;static90 PHIP_STATIC_PERCENT=90;stw5pool4M PHIP_LIB=$PWD/mitsuba_amd/_build/libphip_stw5.so PHIP_POOL=4194304" bash tools/gpu_ab.sh > $o/ab_pool.txt 2>&1
cat $o/ab_pool.txt
