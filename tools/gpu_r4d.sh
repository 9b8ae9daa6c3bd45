#!/bin/bash
# round 4, call d: LDS counters as ds_add / SGPR ballots; full bench line with the D2H inside `value`
set -x
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "cornell or c2 or fused or fuzz or c1 or block_sizes or ragged or progressive or cancel or shards" 2>&1 | tail -5 > gpurun_out/r4d/pytest.txt
cat gpurun_out/r4d/pytest.txt
WORKLOADS="cornell 256" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4d/ab.txt
PHIP_LIB=$PWD/mitsuba_amd/_build/xprof.so SPP=64 python tools/mega_profile.py gpurun_out/r4d/mega_profile.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; tail -c 600 gpurun_out/r4d/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4d/bench.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["workloads"].items(): print(k, v["value"], v.get("value_device_resident"), v["ms_per_step"], v.get("film_d2h_ms"), v["roofline"]["kernel_ms_per_step"] if "roofline" in v else "")
PY
