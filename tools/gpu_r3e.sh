#!/bin/bash
# round 3, GPU call E: SQ counters (lane utilisation, issue, waits) of k_shade with and without the material deal, atrium and glass room
out=gpurun_out/r3e; mkdir -p $out
for sc in atrium glass; do
  for sort in 1 0; do
    PMC_GROUPS=1 SPP=16 PHIP_SHADE_SORT=$sort bash tools/pmc_sq.sh $sc $out/pmc ${sc}_sort$sort
    python tools/pmc_valu.py $out/pmc ${sc}_sort$sort $out/valu_${sc}_sort$sort.json | grep -A 3 "k_shade\|k_rays_w" | head -12
  done
done
rm -f $out/pmc/*agent_info.csv
