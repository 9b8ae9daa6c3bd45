#!/bin/bash
# A/B of library variants on C3 (64 spp) and C4 (128 spp) in one gpurun call: the product first, then every mitsuba_amd/_build/libphip_*.so
run() { echo "== $*"; for s in "atrium 64" "glass 128"; do set -- $s "$@"; env "${@:3}" SPP=$2 python tools/gpu_scenes.py $1 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['Msamples/s'], d['kernel_ms']['trace_kernel_ms'], d['kernel_ms']['shade_kernel_ms'], d['iters'])"; shift 2; done; }
run X=1
for l in mitsuba_amd/_build/libphip_*.so; do run PHIP_LIB=$PWD/$l; done
