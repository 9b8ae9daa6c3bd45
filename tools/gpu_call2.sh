#!/bin/bash
# call 2: k_shade one-round-trip prologue (product) vs the previous build; k_rays_w tuning at 6 waves; parity of the product
mkdir -p gpurun_out
{
echo "== A/B"
WORKLOADS="atrium 64;glass 128" timeout 600 bash tools/gpu_ab.sh
echo "== cornell (fused / wavefront)"
SPP=256 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | cut -c1-400
PHIP_MEGA=0 SPP=256 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | cut -c1-400
PHIP_LIB=$PWD/mitsuba_amd/_build/libphip_prev.so PHIP_MEGA=0 SPP=256 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | cut -c1-400
echo "== parity (product)"
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_bunny.py -m gpu -x -q 2>&1 | tail -5
} > gpurun_out/call2.log 2>&1
