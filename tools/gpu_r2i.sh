#!/bin/bash
# round 2, GPU call I: k_rays_w with one 1024-thread block per CU (800-node LDS cache), spatial-split builder; C2 at full size vs the oracle
out=gpurun_out/r2i; mkdir -p $out
run() { echo "== $*"; env "$@" SPP=64 python tools/gpu_scenes.py atrium 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['Msamples/s'], d['kernel_ms']['trace_kernel_ms'], d['kernel_ms']['shade_kernel_ms'], d['iters'], d['nodes/closest'], d['tris/closest'], d['scene_create_s'])"
 env "$@" SPP=128 python tools/gpu_scenes.py glass 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['Msamples/s'], d['kernel_ms']['trace_kernel_ms'], d['kernel_ms']['shade_kernel_ms'], d['iters'], d['nodes/closest'], d['tris/closest'], d['scene_create_s'])"; }
run X=1
run PHIP_BVH_SPATIAL=0
run PHIP_NODE_CACHE=96
run PHIP_NODE_CACHE=300
run PHIP_NODE_CACHE=585
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "not full_size_against_the_reference" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|rc=|full size|pixel \(|^E " $out/pytest.log | tail -30
