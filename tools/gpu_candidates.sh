out=gpurun_out/r4x; mkdir -p $out
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -m gpu -q -k "sobol or sampler or qmc or halton" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -4 | tee $out/pytest.txt
python - <<'PY' 2>&1 | tee $out/c2_samplers_candidates.txt
import sys, time, json, os
sys.path.insert(0, "tests")
from conftest import sobol_tables, qmc_tables
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm
w=h=1024; spp=256
sc=Scene(S.cornell_box(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=PinnedFilm(w,h)
for name,kw,env in (("sobol",dict(sobol=sobol_tables(w,h)),None),("sobol-nojitterbuf",dict(sobol=sobol_tables(w,h)),"1"),("halton",dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1)),None),("halton-nojitterbuf",dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1)),"1")):
    if env: os.environ["PHIP_NO_JITTER_BUFFER"]=env
    else: os.environ.pop("PHIP_NO_JITTER_BUFFER",None)
    kw.setdefault("flags", A.PHIP_FLAG_KERNEL_TIMING)
    integ.render_into(sc, film.ptr, 4, **kw)
    integ.render_into(sc, film.ptr, spp, **kw)
    t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, **kw); dt=time.perf_counter()-t
    st=integ.stats
    print(json.dumps({"sampler":name,"Msamples/s":round(w*h*spp/1e6/dt,1),"wall_ms":round(dt*1e3,2),"fused_kernel_ms":round(st.fused_kernel_ms,2),"film_ms":round(st.film_kernel_ms,2)}))
PY
SPP=64 python tools/gpu_scenes.py atrium4k 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('atrium4k 64spp', d['Msamples/s'], d['kernel_ms'], d['iters'])" | tee $out/pool.txt
