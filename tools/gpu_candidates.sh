#!/bin/bash
# branch r5-candidates: parity of the fused kernel, then the branch build against main's library (mitsuba_amd/_build/libphip_main.so), interleaved
out=gpurun_out/r4y; mkdir -p $out
b=$PWD/mitsuba_amd/_build
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cornell or c1_config or block_sizes or ragged or overflow or sobol" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3 | tee $out/pytest.txt
row() { env "${@:2}" SPP=256 REPEAT=3 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-8s fused %6.2f ms  film %4.2f  wall %6.1f  %7.1f Msamples/s' % ('$1', d['kernel_ms']['fused_kernel_ms'], d['kernel_ms']['film_kernel_ms'], d['wall_ms'], d['Msamples/s']))"; }
row warm X=1 > /dev/null 2>&1
for i in 1 2 3; do row branch X=1; row main PHIP_LIB=$b/libphip_main.so; done 2>/dev/null | tee $out/ab.txt
