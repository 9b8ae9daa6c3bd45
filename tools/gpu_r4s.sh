#!/bin/bash
# round 4, call s: the round's evidence on the final build: PMC passes of the four workloads (their summaries priced into the bench line that follows),
# bench.py as the driver runs it, rocprofv3 kernel stats of the same command, C2 with the reference's samplers, the phase profile of k_mega, smoke
out=gpurun_out/r4s; mkdir -p $out
PREFIX=r04 STEPS=20 WARMUP=5 bash tools/gpu_profiles.sh r4s 2>&1 | tail -40
python - <<'PY' 2>&1 | tee $out/c2_samplers.txt
import sys, time, json, os
sys.path.insert(0, "tests")
from conftest import sobol_tables, qmc_tables
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm
w=h=1024; spp=256
sc=Scene(S.cornell_box(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=PinnedFilm(w,h)
for name,kw in (("ctr",{}),("sobol",dict(sobol=sobol_tables(w,h))),("halton",dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1))),("ctr",{})):
    kw.setdefault("flags", A.PHIP_FLAG_KERNEL_TIMING)
    integ.render_into(sc, film.ptr, 4, **kw)
    integ.render_into(sc, film.ptr, spp, **kw)
    t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, **kw); dt=time.perf_counter()-t
    st=integ.stats
    print(json.dumps({"sampler":name,"fused":st.fused,"Msamples/s":round(w*h*spp/1e6/dt,1),"wall_ms":round(dt*1e3,2),"fused_kernel_ms":round(st.fused_kernel_ms,2),"film_ms":round(st.film_kernel_ms,2),"d2h_ms":round(st.d2h_ms,3)}))
PY
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -m gpu -q -k "sobol or sampler or qmc or cornell_render" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3 | tee $out/pytest_samplers.txt
python __graft_entry__.py smoke 2>&1 | tail -3 | tee $out/smoke.txt
