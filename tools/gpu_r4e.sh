#!/bin/bash
# round 4, call e: branch-free scene-box clip in k_mega (product vs noclipsel); select-based Wald test in k_rays_w at 6 / 7 / 8 waves
set -x
mkdir -p gpurun_out/r4e
WORKLOADS="cornell 256;atrium 64;glass 128" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4e/ab.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell or c2 or fuzz or atrium or glass or zoo" 2>&1 | tail -5 | tee gpurun_out/r4e/pytest.txt
PHIP_LIB=$PWD/mitsuba_amd/_build/libphip_sel6.so python -m pytest tests/test_gpu_parity.py tests/test_bunny.py -m gpu -x -q -k "atrium or glass or zoo or bunny or soup" 2>&1 | tail -5 | tee gpurun_out/r4e/pytest_sel6.txt
