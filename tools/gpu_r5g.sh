#!/bin/bash
# round 5, GPU call g: C2 with the reference's samplers -- Halton / Hammersley through the multi-digit radical-inverse tables (the digit loops: an experiment
# build with PHIP_RINV_DIGITWISE=1; the A/B row of profiles/r05_gpu_call_g_c2_samplers_rinv_tables.txt was made before the switch left the product), the QMC build of k_mega without the Sobol' row loops (no scratch); sampler parity  -> gpurun_out/r5g/
mkdir -p gpurun_out/r5g
o=gpurun_out/r5g
rm -f mitsuba_amd/_build/libphip_*.so
python - <<'PY' 2>&1 | tee $o/c2_samplers.txt
import sys, time, json, os
sys.path.insert(0, "tests")
from conftest import sobol_tables, qmc_tables
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm
w=h=1024; spp=256
sc=Scene(S.cornell_box(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=PinnedFilm(w,h)
rows=(("ctr",{},{}),("sobol",dict(sobol=sobol_tables(w,h)),{}),("halton",dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1)),{}),
      ("hammersley",dict(sampler=A.PHIP_SAMPLER_HAMMERSLEY, qmc=qmc_tables(-1)),{}),("ctr",{},{}))
for name,kw,env in rows:
    os.environ.update(env)
    kw.setdefault("flags", A.PHIP_FLAG_KERNEL_TIMING)
    integ.render_into(sc, film.ptr, 4, **kw)
    integ.render_into(sc, film.ptr, spp, **kw)
    t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, **kw); dt=time.perf_counter()-t
    st=integ.stats
    print(json.dumps({"sampler":name,"fused":st.fused,"Msamples/s":round(w*h*spp/1e6/dt,1),"wall_ms":round(dt*1e3,2),"fused_kernel_ms":round(st.fused_kernel_ms,2),"film_ms":round(st.film_kernel_ms,2),"d2h_ms":round(st.d2h_ms,3)}))
    for k in env: os.environ.pop(k)
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -m gpu -q -k "sobol or sampler or qmc or halton" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3 | tee $o/pytest_samplers.txt
