#!/bin/bash
# round 4, call h: pass 1 of traverseFlat2 without padding; phase profile
set -x
mkdir -p gpurun_out/r4h
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "cornell or c2 or fused or fuzz or c1" 2>&1 | tail -4 | tee gpurun_out/r4h/pytest.txt
mv mitsuba_amd/_build/xprof.so /tmp/xprof.so
WORKLOADS="cornell 256" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4h/ab.txt
PHIP_LIB=/tmp/xprof.so SPP=64 python tools/mega_profile.py gpurun_out/r4h/mega_profile.json
