#!/bin/bash
# round 2, GPU call E: k_mega at 4 waves (MachineLICM off), shade-kernel occupancy variants, bunny tests, full parity suite
out=gpurun_out/r2e; mkdir -p $out
B=$(pwd)/mitsuba_amd/_build
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
run() { echo "== $1"; shift; env "$@" REPEAT=2 timeout 300 python tools/gpu_scenes.py $SC; }
{
SC=cornell
run "C2 fused 4 waves (product)" SPP=256
run "C2 fused 3 waves, no MLICM" PHIP_LIB=$B/libphip_m3.so SPP=256
run "C2 fused 5 waves (136 B scratch)" PHIP_LIB=$B/libphip_m5.so SPP=256
run "C2 wavefront, product shade (4,4)" PHIP_MEGA=0 SPP=256
run "C2 wavefront, shade (5,6) no MLICM" PHIP_MEGA=0 PHIP_LIB=$B/libphip_s56.so SPP=256
run "C2 wavefront, shade (5,5) no MLICM" PHIP_MEGA=0 PHIP_LIB=$B/libphip_s55.so SPP=256
for SC in atrium glass; do
  if [ $SC = atrium ]; then S=64; else S=128; fi
  run "$SC product" SPP=$S
  run "$SC shade (5,6)" PHIP_LIB=$B/libphip_s56.so SPP=$S
  run "$SC shade (5,5)" PHIP_LIB=$B/libphip_s55.so SPP=$S
done
} > $out/ab.log 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r2e/ab.log'):
    if line.startswith('=='): print(line.strip())
    elif line.startswith('{'):
        d=json.loads(line); k=d["kernel_ms"]; print("   ", d["Msamples/s"], "Msamples/s  fused", k["fused_kernel_ms"], "trace", k["trace_kernel_ms"], "shadow", k["shadow_kernel_ms"], "shade", k["shade_kernel_ms"], "film", k["film_kernel_ms"], "wall", d["wall_ms"])
    else: print(line.strip()[:160])
PY
