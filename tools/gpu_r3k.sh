#!/bin/bash
# round 3, GPU call K: k_mega with its tables addressed as LDS (no flat loads): C2 time + bit identity against the wavefront kernels
SPP=256 REPEAT=3 python tools/gpu_scenes.py cornell | tail -1 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "cornell or fused or c1 or seed" 2>&1 | tail -3
