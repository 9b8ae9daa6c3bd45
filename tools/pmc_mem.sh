#!/bin/bash
# Vector-memory (TA / TCP / TD / TCC) counter passes for one gpu_scenes.py scene -- the resource the big-scene ray kernel is bound by.
# usage (on the GPU box, from the repo root): bash tools/pmc_mem.sh <scene> <outdir> [tag]   [SPP=.. PHIP_LIB=.. in the env]
# Counters only with --kernel-trace (never with sys/hip/hsa tracing, see the gpurun rules); separate --pmc passes.
sc=$1; out=$2; tag=${3:-$sc}; root=$(pwd); mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for c in "TA_FLAT_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum TD_TD_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $root && NOWARM=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out -o ${tag}_m$i --output-format csv -- python tools/gpu_scenes.py $sc > $out/${tag}_m$i.log 2>&1)
done
rm -f $root/$out/*_agent_info.csv
