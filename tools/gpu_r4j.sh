#!/bin/bash
# round 4, call j: the whole -m gpu suite on the final build
set -x
mkdir -p gpurun_out/r4j
python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r4j/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/r4j/smoke.txt
