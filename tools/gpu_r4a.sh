#!/bin/bash
# round 4, call a: traverseFlat2 (packed leaf table + record masks, branch-free Wald test) against the round-3 flat table
set -x
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -m gpu -x -q -k "cornell or c2 or fused or fuzz or c1 or block_sizes or ragged or progressive" 2>&1 | tail -5 > gpurun_out/r4a/pytest.txt
cat gpurun_out/r4a/pytest.txt
WORKLOADS="cornell 256" AB_ENV="flat1 PHIP_NO_FLAT2=1;flat2 X=1;flat1b PHIP_NO_FLAT2=1" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4a/ab.txt
# SQ counters (LDS pipe, issue) of k_mega in both modes
SPP=64 PMC_GROUPS=2 bash tools/pmc_sq.sh cornell gpurun_out/r4a/pmc_flat2 flat2
SPP=64 PMC_GROUPS=2 PHIP_NO_FLAT2=1 bash tools/pmc_sq.sh cornell gpurun_out/r4a/pmc_flat1 flat1
python tools/pmc_valu.py gpurun_out/r4a/pmc_flat2 flat2 gpurun_out/r4a/valu_flat2.json > /dev/null
python tools/pmc_valu.py gpurun_out/r4a/pmc_flat1 flat1 gpurun_out/r4a/valu_flat1.json > /dev/null
python - <<'PY'
import json
for t in ("flat1","flat2"):
    d=json.load(open("gpurun_out/r4a/valu_%s.json"%t))
    for k,v in d.items():
        if k.startswith("k_mega"): print(t,k,json.dumps(v))
PY
