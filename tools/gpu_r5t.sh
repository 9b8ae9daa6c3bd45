#!/bin/bash
# round 5, GPU call t: the whole GPU suite on the final build, then C4 at its full 512 spp against Mitsuba itself with the pinned libm (call u)  -> gpurun_out/r5t/
mkdir -p gpurun_out/r5t
o=gpurun_out/r5t
python -m pytest tests -m gpu -q 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -15 | tee $o/pytest_gpu.txt
python -c "
from mitsuba_amd import _ffi
print('build id', _ffi.built_id(_ffi.LIB))" | tee -a $o/pytest_gpu.txt
LD_LIBRARY_PATH=$PWD/oracle/_ref/pinned_libm:$LD_LIBRARY_PATH timeout 1000 python tools/fullsize_vs_reference.py $o/c4_fullsize_gpu_vs_reference.json C4full 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -3
