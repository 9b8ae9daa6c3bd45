#!/bin/bash
# final evidence of the round on ONE box: PMC passes + bench + kernel stats (tools/gpu_profiles.sh), then the whole -m gpu suite and smoke()
mkdir -p gpurun_out/r3g
PREFIX=r03b bash tools/gpu_profiles.sh r3g > gpurun_out/r3g/profiles_run.log 2>&1; tail -22 gpurun_out/r3g/profiles_run.log | cut -c1-300
timeout 1100 python -m pytest tests/ -m gpu -q > gpurun_out/r3g/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3g/pytest_gpu.txt
grep -E "passed|failed|error" gpurun_out/r3g/pytest_gpu.txt | tail -3
python __graft_entry__.py smoke > gpurun_out/r3g/smoke.txt 2>&1; tail -1 gpurun_out/r3g/smoke.txt | cut -c1-200
