#!/bin/bash
# round 4, call p: the dealt triangle rounds as the product (threshold 32, six stack entries in LDS): parity of the big-scene tests, then the sweep
# around it (flat = the loop it replaces, thresholds 24 / 40, refill at 8 / 24 idle lanes, seven stack entries with a 32-node cache)
out=gpurun_out/r4p; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py tests/test_gpu_direct.py -m gpu -q 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -6 | tee $out/pytest.txt
WORKLOADS="atrium 64;glass 128" bash tools/gpu_ab.sh 2>&1 | tee $out/ab.txt
