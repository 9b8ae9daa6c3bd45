#!/bin/bash
# round 4, call w (measurement only, environment switches, the product library): pool size around the dealt ray kernel -- C3 (default 16 M slots) and the C5 slice (32 M)
out=gpurun_out/r4w; mkdir -p $out
AB_ENV="${AB_ENV:-pool8M PHIP_POOL=8388608;pool16M PHIP_POOL=16777216;pool32M PHIP_POOL=33554432;pool64M PHIP_POOL=67108864}" WORKLOADS="${WORKLOADS:-atrium 64;atrium4k 64}" bash tools/gpu_ab.sh 2>&1 | tee $out/${TAG:-ab}.txt
