#!/bin/bash
# round 2, GPU call A: parity suite on the new build, fused-kernel A/B on C2, SQ counter baselines at full size
out=gpurun_out/r2a; mkdir -p $out
B=$(pwd)/mitsuba_amd/_build
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
{
echo "== C2 wavefront (PHIP_MEGA=0)"; PHIP_MEGA=0 SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
echo "== C2 fused, 3 waves/SIMD (product)"; SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
echo "== C2 fused, 2 blocks per CU"; PHIP_MEGA_BLOCKS=2 SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
echo "== C2 fused, 1 block per CU"; PHIP_MEGA_BLOCKS=1 SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
echo "== C2 fused, build with 2 waves/SIMD"; PHIP_LIB=$B/libphip_mw2.so SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
echo "== C2 fused, build with 4 waves/SIMD (scratch)"; PHIP_LIB=$B/libphip_mw4.so SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
echo "== C3"; SPP=64 REPEAT=2 timeout 300 python tools/gpu_scenes.py atrium
echo "== C4"; SPP=512 timeout 300 python tools/gpu_scenes.py glass
} > $out/ab.log 2>&1
cat $out/ab.log
SPP=256 bash tools/pmc_sq.sh cornell $out/pmc c2_fused
PHIP_MEGA=0 PMC_GROUPS=1 SPP=256 bash tools/pmc_sq.sh cornell $out/pmc c2_wavefront
SPP=64 bash tools/pmc_sq.sh atrium $out/pmc c3
python tools/pmc_valu.py $out/pmc c2_fused $out/pmc_c2_fused.json > /dev/null
python tools/pmc_valu.py $out/pmc c2_wavefront $out/pmc_c2_wavefront.json > /dev/null
python tools/pmc_valu.py $out/pmc c3 $out/pmc_c3.json > /dev/null
rm -f $out/pmc/*_agent_info.csv
grep -h "valu_issue_frac\|lane_util\|valu_frac\|\"k_\|avg_launch_us\|wait_frac\|waves_per_simd" $out/pmc_*.json | head -80
