#!/bin/bash
# round 4, call b: Wald-loop prefetch variant, phase profile of the flat2 kernel, film D2H paths
set -x
mkdir -p gpurun_out/r4b
WORKLOADS="cornell 256" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4b/ab.txt
PHIP_LIB=$PWD/mitsuba_amd/_build/libphip_prof.so SPP=64 python tools/mega_profile.py gpurun_out/r4b/mega_profile_flat2.json
python - <<'PY' 2>&1 | tee gpurun_out/r4b/d2h.txt
import time, numpy as np, ctypes as C
from mitsuba_amd import _ffi, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm, PinnedFilm
w=h=1024
sc=Scene(S.cornell_box(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1)
pf=PinnedFilm(w,h); block=np.zeros((h,w,5),np.float32)
for name,ptr in (("pinned",pf.ptr),("pageable",block.ctypes.data)):
    for i in range(3):
        t=time.perf_counter(); integ.render_into(sc,ptr,16); dt=time.perf_counter()-t
        print(name,i,"render_ms %.2f d2h_ms %.3f wall %.2f"%(integ.stats.render_ms,integ.stats.d2h_ms,dt*1e3))
assert (pf.storage==block).all()
print("pinned == pageable frame: ok")
PY
