#!/bin/bash
# round 5, last call: parity of the final build (the whole tests/test_gpu_parity.py), then the round's evidence (tools/gpu_r5s.sh)
mkdir -p gpurun_out/r5s
timeout 600 python -m pytest tests/test_gpu_parity.py -q 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -6 | tee gpurun_out/r5s/pytest_parity_final.txt
bash tools/gpu_r5s.sh
