#!/bin/bash
# A/B of k_mega variants on C2 (one gpurun call): PHIP_LIB=<variant> per line
run() { echo "== $*"; env "$@" SPP=256 REPEAT=3 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['Msamples/s'], d['kernel_ms']['fused_kernel_ms'], d['kernel_ms']['film_kernel_ms'], d['nodes/closest'], d['tris/closest'])"; }
run X=1
for l in mitsuba_amd/_build/libphip_*.so; do run PHIP_LIB=$PWD/$l; done
