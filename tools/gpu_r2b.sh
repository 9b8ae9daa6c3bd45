#!/bin/bash
# round 2, GPU call B: wide BVH parity + A/B on C3 / C4
out=gpurun_out/r2b; mkdir -p $out
B=$(pwd)/mitsuba_amd/_build
timeout 900 python -m pytest tests -m gpu -x -q -k "raycast or atrium or glass or c3 or c4 or fullsize or multi_device or axis" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
{
echo "== C3 BVH4 (PHIP_WIDE=0)"; PHIP_WIDE=0 SPP=64 REPEAT=2 timeout 300 python tools/gpu_scenes.py atrium
echo "== C3 wide, 5 waves (scratch 64 B)"; SPP=64 REPEAT=2 timeout 300 python tools/gpu_scenes.py atrium
echo "== C3 wide, 4 waves"; PHIP_LIB=$B/libphip_ww4.so SPP=64 REPEAT=2 timeout 300 python tools/gpu_scenes.py atrium
echo "== C3 wide, node cache 0"; PHIP_NODE_CACHE=0 SPP=64 REPEAT=2 timeout 300 python tools/gpu_scenes.py atrium
echo "== C4 BVH4"; PHIP_WIDE=0 SPP=128 timeout 300 python tools/gpu_scenes.py glass
echo "== C4 wide"; SPP=128 timeout 300 python tools/gpu_scenes.py glass
echo "== C4 wide 4 waves"; PHIP_LIB=$B/libphip_ww4.so SPP=128 timeout 300 python tools/gpu_scenes.py glass
} > $out/ab.log 2>&1
cat $out/ab.log
PMC_GROUPS=1 SPP=64 bash tools/pmc_sq.sh atrium $out/pmc c3_wide
python tools/pmc_valu.py $out/pmc c3_wide $out/pmc_c3_wide.json > /dev/null
rm -f $out/pmc/*_agent_info.csv
grep -h "valu_issue_frac\|lane_util\|valu_frac\|\"k_\|avg_launch_us\|wait_frac\|waves_per_simd" $out/pmc_*.json | head -60
