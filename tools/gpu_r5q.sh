#!/bin/bash
# round 5, GPU call q: the order of the class deal in k_mega<MM_ALL> -- copper, diffuse, glass, idle (the product: the two expensive models in different waves)
# against copper, glass, diffuse, idle (tools/build_variant.sh order0, -DMEGA_DEAL_ORDER=0); then the whole GPU suite on the product   -> gpurun_out/r5q/
mkdir -p gpurun_out/r5q
o=gpurun_out/r5q
WORKLOADS="cmixed 256" bash tools/gpu_ab.sh 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tee $o/mega_deal_order_ab.txt
rm -f mitsuba_amd/_build/libphip_*.so
python -m pytest tests -m gpu -q 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -15 | tee $o/pytest_gpu.txt
python -c "
from mitsuba_amd import _ffi
print('build id', _ffi.built_id(_ffi.LIB))" | tee -a $o/pytest_gpu.txt
