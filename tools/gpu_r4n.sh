#!/bin/bash
# round 4, call n: k_rays_w with the triangle tests of an iteration dealt over the wave (WIDE_DEAL): parity of the variant, then A/B on C3 / C4 (128 spp)
out=gpurun_out/r4n; mkdir -p $out
b=$PWD/mitsuba_amd/_build
PHIP_LIB=$b/libphip_t40.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "atrium or glass or zoo or fuzz or c3 or c4 or c5 or environment or envmap or textures or large_emitter" 2>&1 | tail -6 | tee $out/pytest_t40.txt
WORKLOADS="atrium 64;glass 128" bash tools/gpu_ab.sh 2>&1 | tee $out/ab.txt
