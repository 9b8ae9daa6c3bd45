#!/bin/bash
# round 5, GPU call o: where the 2 x cost of the mixed Cornell box comes from -- (1) the all-diffuse box on the kernel that knows all three BSDF models
# (experiment build, PHIP_SHADE_GENERIC=1: the price of the code's generality without any divergence), (2) the VALU counters of the mixed box with the block's
# paths dealt by BSDF model, (3) ... of (1)   -> gpurun_out/r5o/
mkdir -p gpurun_out/r5o
o=gpurun_out/r5o
exp=$PWD/mitsuba_amd/_build/libphip_exp.so
AB_ENV="generic PHIP_LIB=$exp PHIP_SHADE_GENERIC=1;exp PHIP_LIB=$exp" WORKLOADS="cornell 256;cmixed 256" bash tools/gpu_ab.sh 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids\|^exp .*libphip_exp" | tee $o/generic_kernel_ab.txt
PMC_GROUPS=1 SPP=256 bash tools/pmc_sq.sh cmixed $o/pmc cmixed_deal
python tools/pmc_valu.py $o/pmc cmixed_deal $o/valu_cmixed_deal.json | tail -2
PHIP_LIB=$exp PHIP_SHADE_GENERIC=1 PMC_GROUPS=1 SPP=256 bash tools/pmc_sq.sh cornell $o/pmc cornell_generic
python tools/pmc_valu.py $o/pmc cornell_generic $o/valu_cornell_generic.json | tail -2
rm -rf $o/pmc/*_agent_info.csv $o/pmc/*kernel_trace.csv
