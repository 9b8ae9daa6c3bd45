#!/bin/bash
# SQ counter passes (issue / wait / lane utilisation) for one gpu_scenes.py scene.
# usage (on the GPU box, from the repo root): bash tools/pmc_sq.sh <scene> <outdir> [tag]   [SPP=.. PHIP_MEGA=.. in the env]
# Counters only with --kernel-trace (never with sys/hip/hsa tracing, see the gpurun rules).
sc=$1; out=$2; tag=${3:-$sc}; root=$(pwd); mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
         "SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  [ -n "$PMC_GROUPS" ] && [ $i -gt $PMC_GROUPS ] && break
  (cd $root && NOWARM=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out -o ${tag}_g$i --output-format csv -- python tools/gpu_scenes.py $sc > $out/${tag}_g$i.log 2>&1)
done
