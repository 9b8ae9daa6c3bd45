#!/bin/bash
# round 2, GPU call D: wide BVH with the SAH-optimal collapse, typed LDS node cache, 3-load records -- parity + A/B on C3 / C4
out=gpurun_out/r2d; mkdir -p $out
B=$(pwd)/mitsuba_amd/_build
timeout 900 python -m pytest tests -m gpu -x -q -k "raycast or atrium or glass or c3 or c4 or fullsize or axis" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
run() { echo "== $1"; shift; env "$@" REPEAT=2 timeout 300 python tools/gpu_scenes.py $SC; }
{
for SC in atrium glass; do
  if [ $SC = atrium ]; then S=64; else S=128; fi
  run "$SC BVH4" PHIP_WIDE=0 SPP=$S
  run "$SC wide (product: 5 waves, typed cache 96, refill 16)" SPP=$S
  run "$SC wide, no LDS node cache" PHIP_NODE_CACHE=0 SPP=$S
  run "$SC wide, 4 waves" PHIP_LIB=$B/libphip_ww4.so SPP=$S
  run "$SC wide, refill 8" PHIP_LIB=$B/libphip_r8.so SPP=$S
  run "$SC wide, flat loads for cached nodes" PHIP_LIB=$B/libphip_nt.so SPP=$S
done
} > $out/ab.log 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r2d/ab.log'):
    if line.startswith('=='): print(line.strip())
    elif line.startswith('{'):
        d=json.loads(line); print("   ", d["Msamples/s"], "Msamples/s  trace", d["kernel_ms"]["trace_kernel_ms"], "shade", d["kernel_ms"]["shade_kernel_ms"], "nodes/closest", d["nodes/closest"], "tris", d["tris/closest"], "n_nodes", d["accel"]["n_nodes"])
    else: print(line.strip()[:160])
PY
