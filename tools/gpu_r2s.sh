#!/bin/bash
# round 2, GPU call S (measurement after the flat leaf table of k_mega): bench.py, kernel stats, PMC traffic + VALU of C2
out=gpurun_out/r2s; mkdir -p $out
root=$(pwd)
timeout 900 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], {k:v['value'] for k,v in d['workloads'].items()}, d['roofline']['kernel_ms_per_step'], d['cpu_baseline']['value'])"
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $root/$out/prof -o bench --output-format csv -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/$out/prof_bench.json 2> $root/$out/prof.err)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1); python tools/rocprof_summary.py $f "bench.py --steps 2 --warmup 1 (C2 + C3 + C4 + C5 slice), MI355X" > $out/kernel_stats.md 2>&1; head -14 $out/kernel_stats.md
for w in "cornell 256 cornell_1024x1024_256spp"; do
  set -- $w
  timeout 900 python tools/pmc_traffic.py $1 $out/traffic_$3.json $2 $3 2>&1 | tail -1
  PMC_GROUPS=1 SPP=$2 bash tools/pmc_sq.sh $1 $out/pmc $3
  python tools/pmc_valu.py $out/pmc $3 $out/valu_$3.json | tail -3
done
rm -rf $out/pmc/*_agent_info.csv $out/prof/*agent_info* $out/prof/*kernel_trace.csv gpurun_out/pmc_traffic
