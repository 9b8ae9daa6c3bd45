#!/bin/bash
# round 4, call m: k_mega's dealt test loop with 1 / 2 / 4 pairs per lane and step (BAL_ILP), work counters in registers (MEGA_COUNT_REGS); interleaved rows
out=gpurun_out/r4m; mkdir -p $out
b=$PWD/mitsuba_amd/_build
row() { env "${@:2}" SPP=256 REPEAT=3 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-9s fused %6.2f ms  film %4.2f  wall %6.1f  %7.1f Msamples/s' % ('$1', d['kernel_ms']['fused_kernel_ms'], d['kernel_ms']['film_kernel_ms'], d['wall_ms'], d['Msamples/s']))"; }
row warmup X=1 > /dev/null 2>&1
for i in 1 2 3; do row product X=1; for v in ilp2 ilp4 ilp2regs regs; do row $v PHIP_LIB=$b/libphip_$v.so; done; done 2>/dev/null | tee $out/ilp_ab.txt
for v in ilp2 ilp4 ilp2regs; do PHIP_LIB=$b/libphip_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cornell or c1_config or block_sizes or ragged or sobol" 2>&1 | tail -1; done | tee $out/pytest_variants.txt
