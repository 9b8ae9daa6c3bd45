#!/bin/bash
# SQ counter passes (issue / wait / lane utilisation) for one gpu_scenes.py scene.
# usage (on the GPU box, from the repo root): bash tools/pmc_sq.sh <scene> <outdir>   [SPP=.. in the env]
# Counters only with --kernel-trace (never with sys/hip/hsa tracing, see the gpurun rules).
sc=$1; out=$2; root=$(pwd); mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
         "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1))
  (cd $root && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out -o ${sc}_g$i --output-format csv -- python tools/gpu_scenes.py $sc > $out/${sc}_g$i.log 2>&1)
done
