#!/bin/bash
# Evidence run on the GPU box (one gpurun call): the PMC passes (HBM traffic, SQ / VALU, TA / TCP / TD / TCC) of every workload of the bench
# line FIRST -- their summaries are copied to profiles/<prefix>_*.json on the box, so that the bench line that follows prices its kernels with
# the counters of the SAME build --, then bench.py as the driver runs it, then rocprofv3 kernel stats of the same command.
#   PREFIX=r03b bash tools/gpu_profiles.sh <tag>   ->  gpurun_out/<tag>/{bench.json, kernel_stats.md, traffic_*.json, valu_*.json, tcp_*.json}
# The summaries are copied into profiles/ (tracked) by hand afterwards: profiles/<prefix>_*.
tag=${1:-prof}; out=gpurun_out/$tag; mkdir -p $out; root=$(pwd); prefix=${PREFIX:-r03b}
if [ -z "$NO_PMC" ]; then
for w in "cornell 256 cornell_1024x1024_256spp" "atrium 64 atrium_1920x1080_64spp_md8" "glass 512 glassroom_1920x1080_512spp_md16" "atrium4k 64 atrium_3840x2160_64spp_md8" "cmixed 256 cornell_mixed_1024x1024_256spp" "sph1k 64 cornell_spheres_1k_1024x1024_64spp" "sph1kd 64 cornell_spheres_1k_diffuse_1024x1024_64spp"; do
  set -- $w
  timeout 900 python tools/pmc_traffic.py $1 $out/traffic_$3.json $2 $3 2>&1 | tail -1
  PMC_GROUPS=1 SPP=$2 bash tools/pmc_sq.sh $1 $out/pmc $3
  python tools/pmc_valu.py $out/pmc $3 $out/valu_$3.json | tail -2
  SPP=$2 bash tools/pmc_mem.sh $1 $out/pmc $3
  python tools/pmc_tcp.py $out/pmc $3 $out/tcp_$3.json profiles/r03_vmem_roof.json | head -3
  for k in traffic valu tcp; do [ -s $out/${k}_$3.json ] && cp $out/${k}_$3.json profiles/${prefix}_${k}_$3.json; done
done
fi
timeout 1500 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open('$out/bench.json'))
print(d['value'], d['ms_per_step'], {k: v['value'] for k, v in d.get('workloads', {}).items()}, d['roofline']['kernel_ms_per_step'], d['roofline']['bound'], d['roofline'].get('fractions'), (d.get('cpu_baseline') or {}).get('value'))
PY
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $root/$out/prof -o bench --output-format csv -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/$out/prof_bench.json 2> $root/$out/prof.err)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1); python tools/rocprof_summary.py $f "bench.py --steps 2 --warmup 1 --no-cpu-baseline (C2 + C3 + C4 + C5 slice + mixed Cornell box + C2 with direct + the two sphere boxes), MI355X" > $out/kernel_stats.md 2>&1; head -16 $out/kernel_stats.md
rm -rf $out/pmc/*_agent_info.csv $out/pmc/*kernel_trace.csv $out/prof gpurun_out/pmc_traffic
