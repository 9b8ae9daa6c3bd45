#!/bin/bash
# round 5, GPU call n: k_mega<MM_ALL> with the block's paths dealt to its lanes by BSDF model (MEGA_CLASS_DEAL) against the build without it
# (MEGA_FLAGS="-mllvm -disable-machine-licm -DMEGA_CLASS_DEAL=0" tools/build_variant.sh nodeal) on the mixed Cornell box and C2; parity   -> gpurun_out/r5n/
mkdir -p gpurun_out/r5n
o=gpurun_out/r5n
WORKLOADS="cmixed 256;cornell 256" bash tools/gpu_ab.sh 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tee $o/mega_class_deal_ab.txt
for lib in mitsuba_amd/_build/libphip.so mitsuba_amd/_build/libphip_nodeal.so; do
PHIP_LIB=$PWD/$lib python - <<'PY' 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tee -a $o/mega_class_deal_ab.txt
import sys, time, json, os
sys.path.insert(0, "tests")
from conftest import sobol_tables, qmc_tables
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, VolPathSimpleHIP, PinnedFilm
w=h=1024; spp=256
sc=Scene(S.cornell_mixed(w,h,_ffi.gaussian_filter()).desc()); film=PinnedFilm(w,h)
for name,integ,kw in (("sobol",PathHIP(maxDepth=-1),dict(sobol=sobol_tables(w,h))),("halton",PathHIP(maxDepth=-1),dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1))),
                      ("volpath_simple",VolPathSimpleHIP(maxDepth=-1),{}),("md8",PathHIP(maxDepth=8),{})):
    integ.render_into(sc, film.ptr, 4, **kw)
    integ.render_into(sc, film.ptr, spp, **kw)
    t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, **kw); dt=time.perf_counter()-t
    print(os.path.basename(os.environ["PHIP_LIB"]), "cmixed", name, "fused", integ.stats.fused, round(w*h*spp/1e6/dt,1), "Msamples/s")
PY
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -4 | tee $o/pytest_parity.txt
