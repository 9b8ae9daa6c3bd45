#!/bin/bash
# round 4, call g: film splat + merge kernels, direct + QMC, sobol loops with batched reads; shard balance of the C5 slice
set -x
mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -m gpu -x -q -k "cornell or c1 or block_sizes or ragged or shards or filter_widths or sobol or halton or direct or ld_sampler or empty or seed" 2>&1 | tail -8 | tee gpurun_out/r4g/pytest.txt
python - <<'PY' 2>&1 | tee gpurun_out/r4g/c2_samplers.txt
import sys, time, json, os
sys.path.insert(0, "tests")
from conftest import sobol_tables, qmc_tables
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm
w=h=1024; spp=256
sc=Scene(S.cornell_box(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=PinnedFilm(w,h)
for name,kw in (("ctr",{}),("sobol",dict(sobol=sobol_tables(w,h))),("halton",dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1)))):
    kw.setdefault("flags", A.PHIP_FLAG_KERNEL_TIMING)
    integ.render_into(sc, film.ptr, 4, **kw)
    t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, **kw); dt=time.perf_counter()-t
    st=integ.stats
    print(json.dumps({"sampler":name,"fused":st.fused,"Msamples/s":round(w*h*spp/1e6/dt,1),"wall_ms":round(dt*1e3,2),"fused_kernel_ms":round(st.fused_kernel_ms,2),"film_ms":round(st.film_kernel_ms,2),"d2h_ms":round(st.d2h_ms,3)}))
os.environ["PHIP_FILM_GATHER"]="1"
integ.render_into(sc, film.ptr, spp, flags=A.PHIP_FLAG_KERNEL_TIMING); st=integ.stats
print(json.dumps({"sampler":"ctr, film gather (round 3)","film_ms":round(st.film_kernel_ms,2)}))
PY
WORKLOADS="atrium 64;glass 128" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4g/ab.txt
python tools/shard_balance.py gpurun_out/r4g/shard_balance.json
