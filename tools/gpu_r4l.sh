#!/bin/bash
# round 4, call l: (1) the centre / half-extent box table against the plane table, interleaved (product = MEGA_FLAT_CH 1, noch = 0; both with the
# dealt Wald tests) -- call k's single rows drifted by 4 ms with the chip's temperature; (2) SQ counters of the new k_mega (VALU issue, lane utilisation)
set -x
out=gpurun_out/r4l; mkdir -p $out
b=$PWD/mitsuba_amd/_build
row() { env "${@:2}" SPP=256 REPEAT=3 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-8s fused %6.2f ms  film %4.2f  wall %6.1f  %7.1f Msamples/s' % ('$1', d['kernel_ms']['fused_kernel_ms'], d['kernel_ms']['film_kernel_ms'], d['wall_ms'], d['Msamples/s']))"; }
for i in 1 2 3 4; do row ch X=1; row noch PHIP_LIB=$b/libphip_noch.so; done 2>/dev/null | tee $out/ch_ab.txt
PMC_GROUPS=2 SPP=64 bash tools/pmc_sq.sh cornell $out/pmc cornell64
python tools/pmc_valu.py $out/pmc cornell64 $out/sq_cornell_64spp_balanced.json | tail -3
rm -rf $out/pmc/*_agent_info.csv $out/pmc/*kernel_trace.csv
