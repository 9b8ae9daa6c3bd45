#!/bin/bash
# round-3 (second session) call 1: k_rays_w flat loop at 4..8 waves per SIMD, parity of the product build (6 waves)
mkdir -p gpurun_out
{
echo "== parity (product: flat loop, 6 waves)"
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_bunny.py tests/test_gpu_round2.py -m gpu -x -q -k "raycast or atrium or glass_room or bunny or soup or zero or axis" 2>&1 | tail -5
echo "== A/B"
WORKLOADS="atrium 64;glass 128" timeout 600 bash tools/gpu_ab.sh
} > gpurun_out/call1.log 2>&1
