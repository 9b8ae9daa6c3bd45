#!/bin/bash
# round 2, GPU call F: bench.py as the driver runs it, rocprofv3 kernel stats of the same command, PMC traffic + VALU counters per workload
out=gpurun_out/r2f; mkdir -p $out
root=$(pwd)
timeout 900 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 3000 $out/bench.json; tail -3 $out/bench.err
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $root/$out/prof -o bench --output-format csv -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/$out/prof_bench.json 2> $root/$out/prof.err)
ls $out/prof | head
f=$(find $out/prof -name "*kernel_trace.csv" | head -1); python tools/rocprof_summary.py $f "bench.py --steps 2 --warmup 1 (C2 + C3 + C4 + C5 slice), MI355X" > $out/kernel_stats.md 2>&1
head -30 $out/kernel_stats.md
for w in "cornell 256 cornell_1024x1024_256spp" "atrium 64 atrium_1920x1080_64spp_md8" "glass 512 glassroom_1920x1080_512spp_md16"; do
  set -- $w
  timeout 900 python tools/pmc_traffic.py $1 $out/traffic_$3.json $2 $3 2>&1 | tail -2
  PMC_GROUPS=1 SPP=$2 bash tools/pmc_sq.sh $1 $out/pmc $3
  python tools/pmc_valu.py $out/pmc $3 $out/valu_$3.json > /dev/null
done
rm -f $out/pmc/*_agent_info.csv $out/prof/*agent_info*
ls -la $out
