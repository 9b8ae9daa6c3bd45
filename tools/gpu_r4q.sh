#!/bin/bash
# round 4, call q: Sobol' numbers from byte tables (dv_math.h: SobolTab::matBt) against the row loops (PHIP_SOBOL_BITWISE=1): parity of every test that
# draws from the sobol sampler, then C2 with the counter stream / sobol (both forms) / halton
out=gpurun_out/r4q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_dropin.py -m gpu -q -k "sobol or qmc or sampler" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5 | tee $out/pytest.txt
python - <<'PY' 2>&1 | tee $out/c2_samplers.txt
import sys, time, json, os
sys.path.insert(0, "tests")
from conftest import sobol_tables, qmc_tables
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm
w=h=1024; spp=256
sc=Scene(S.cornell_box(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=PinnedFilm(w,h)
for name,kw,env in (("ctr",{},None),("sobol",dict(sobol=sobol_tables(w,h)),None),("sobol-bitwise",dict(sobol=sobol_tables(w,h)),"1"),("halton",dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1)),None),("sobol",dict(sobol=sobol_tables(w,h)),None)):
    if env: os.environ["PHIP_SOBOL_BITWISE"]=env
    else: os.environ.pop("PHIP_SOBOL_BITWISE",None)
    kw.setdefault("flags", A.PHIP_FLAG_KERNEL_TIMING)
    integ.render_into(sc, film.ptr, 4, **kw)
    t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, **kw); dt=time.perf_counter()-t
    st=integ.stats
    print(json.dumps({"sampler":name,"fused":st.fused,"Msamples/s":round(w*h*spp/1e6/dt,1),"wall_ms":round(dt*1e3,2),"fused_kernel_ms":round(st.fused_kernel_ms,2),"film_ms":round(st.film_kernel_ms,2),"d2h_ms":round(st.d2h_ms,3)}))
PY
