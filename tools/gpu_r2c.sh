#!/bin/bash
# round 2, GPU call C: memory-pipeline counters of k_rays_w (C3) + SAH parameter sweep for the fused kernel (C2)
out=gpurun_out/r2c; mkdir -p $out/pmc; root=$(pwd)
{
for cfg in "4 1.0" "8 1.0" "8 2.0" "8 4.0" "6 2.0" "8 8.0" "2 1.0"; do
  set -- $cfg
  echo "== C2 fused MAXLEAF=$1 CTRAV=$2"; PHIP_BVH_MAXLEAF=$1 PHIP_BVH_CTRAV=$2 SPP=256 REPEAT=2 timeout 300 python tools/gpu_scenes.py cornell
done
} > $out/sah.log 2>&1
grep -o '== .*\|"Msamples/s": [0-9.]*\|"n_nodes": [0-9]*\|"nodes/closest": [0-9.]*\|"tris/closest": [0-9.]*\|"fused": [0-9]' $out/sah.log | paste - - - - - - 
cd /tmp; export TMPDIR=/tmp
i=0
for c in "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TD_TD_BUSY_sum" \
         "TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum TA_FLAT_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  (cd $root && SPP=16 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc -o mem_g$i --output-format csv -- python tools/gpu_scenes.py atrium > $out/pmc/mem_g$i.log 2>&1; tail -2 $out/pmc/mem_g$i.log | cut -c1-300)
done
cd $root; rm -f $out/pmc/*_agent_info.csv
python tools/pmc_summary.py $out/pmc mem | grep -A40 "## k_rays_w" | head -60
