#!/bin/bash
# evidence call A: the whole -m gpu suite and smoke() on the final build
mkdir -p gpurun_out/r3f
timeout 1100 python -m pytest tests/ -m gpu -q > gpurun_out/r3f/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest_gpu.txt
grep -E "passed|failed|error" gpurun_out/r3f/pytest_gpu.txt | tail -3
python __graft_entry__.py smoke > gpurun_out/r3f/smoke.txt 2>&1; tail -2 gpurun_out/r3f/smoke.txt
