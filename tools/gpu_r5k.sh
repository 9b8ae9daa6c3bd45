#!/bin/bash
# round 5, GPU call k: the fused kernel on all three leaf BSDF models (k_mega<MM_ALL>, packed leaf table) -- the mixed Cornell box on k_mega / k_shade_trace
# (PHIP_FLAG_NO_MEGA through the harness) / the three-kernel iterations; parity   -> gpurun_out/r5k/
mkdir -p gpurun_out/r5k
o=gpurun_out/r5k
python - <<'PY' 2>&1 | tee $o/cmixed_paths.txt
import sys, time, json
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm
w=h=1024; spp=256
for scene in ("cornell_mixed", "cornell_box"):
    sc=Scene(getattr(S, scene)(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=PinnedFilm(w,h)
    for name, fl in (("k_mega", 0), ("k_shade_trace", A.PHIP_FLAG_NO_MEGA), ("three kernels", A.PHIP_FLAG_NO_FUSED), ("k_mega", 0)):
        integ.render_into(sc, film.ptr, 4, flags=fl)
        integ.render_into(sc, film.ptr, spp, flags=fl | A.PHIP_FLAG_KERNEL_TIMING)
        t=time.perf_counter(); integ.render_into(sc, film.ptr, spp, flags=fl | A.PHIP_FLAG_KERNEL_TIMING); dt=time.perf_counter()-t
        st=integ.stats
        print(json.dumps({"scene": scene, "path": name, "fused": st.fused, "vertex_traced": st.vertex_traced, "Msamples/s": round(w*h*spp/1e6/dt,1), "wall_ms": round(dt*1e3,2),
                          "fused_ms": round(st.fused_kernel_ms,2), "shade_ms": round(st.shade_kernel_ms,2), "rays_ms": round(st.trace_kernel_ms+st.shadow_kernel_ms,2), "film_ms": round(st.film_kernel_ms,2),
                          "mean_path_length": round(st.path_vertices/st.samples,3), "iterations": st.iterations}))
    sc.close()
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -4 | tee $o/pytest_parity.txt
