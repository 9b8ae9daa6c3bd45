#!/bin/bash
# round 5, GPU call p: MEGA_CLASS_DEAL with the exchange skipped in passes whose copper / glass vertices already lie in as few waves as they fill
# (MEGA_DEAL_SKIP, the product) against the exchange in every pass that has any (tools/build_variant.sh noskip)   -> gpurun_out/r5p/
mkdir -p gpurun_out/r5p
o=gpurun_out/r5p
WORKLOADS="cmixed 256" bash tools/gpu_ab.sh 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tee $o/mega_deal_skip_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "mixed or volpath or records" 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -4 | tee $o/pytest_parity.txt
