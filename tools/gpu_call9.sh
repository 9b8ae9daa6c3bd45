#!/bin/bash
# call 10 (measurement only): is the seventh wave or its smaller LDS stack / node cache what loses?  5, 6, 7 waves per SIMD with the SAME 9-entry stack and 48-node cache
# to a second, dependent one (dl2) and to 24 more VALU instructions per node step (dv24) -- the experiments that read "latency-bound" at four waves
mkdir -p gpurun_out
WORKLOADS="atrium 64;glass 128" timeout 600 bash tools/gpu_ab.sh > gpurun_out/call10.log 2>&1
