"""Runs the BASELINE configs on the GPU (reduced spp by env) and prints throughput + counters."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm

ft = _ffi.gaussian_filter()
cfgs = {"cornell": ("cornell_box", 1024, 1024, 64, -1), "atrium": ("atrium", 1920, 1080, 16, 8), "glass": ("glass_room", 1920, 1080, 32, 16),
        "atrium4k": ("atrium", 3840, 2160, 16, 8),
        "cmixed": ("cornell_mixed", 1024, 1024, 64, -1),                       # the Cornell box with a copper and a glass block (wavefront kernels)
        "c42": ("cornell_box", 1024, 1024, 64, -1, {"extra_blocks": 1}),       # 42 / 62 Wald records: the two-word record masks of the fused kernel
        "c62": ("cornell_box", 1024, 1024, 64, -1, {"extra_blocks": 3}),
        "c82": ("cornell_box", 1024, 1024, 64, -1, {"extra_blocks": 5}),       # 82 records: past the packed leaf table -- the 8-wide tree (round 6; before: k_mega's BVH4 walk in LDS)
        "c92": ("cornell_box", 1024, 1024, 64, -1, {"extra_blocks": 6}),
        # the mid-sized scenes: the Cornell box with a glass and a copper sphere (or two diffuse ones) of 1 k / 4.5 k / 18 k triangles: a tree that lives in L2
        "cglass": ("cornell_box", 1024, 1024, 64, -1, {"tall_bsdf": lambda b: b.dielectric()}),            # the box with a glass block only / a copper block only
        "ccopper": ("cornell_box", 1024, 1024, 64, -1, {"short_bsdf": lambda b: b.twosided(b.roughconductor(S.CU_ETA, S.CU_K, alpha=0.1))}),
        "sph1k": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 24, "nlat": 12}), "sph5k": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 48, "nlat": 24}),
        "sph18k": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 96, "nlat": 48}),
        "sph1kd": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 24, "nlat": 12, "materials": False}),
        "sph18kd": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 96, "nlat": 48, "materials": False})}
for key in sys.argv[1:] or cfgs:
    name, w, h, spp, md = cfgs[key][:5]
    spp = int(os.environ.get("SPP", spp))
    sb = getattr(S, name)(w, h, ft, **(cfgs[key][5] if len(cfgs[key]) > 5 else {}))
    t = time.time(); sc = Scene(sb.desc()); tb = time.time() - t
    if os.environ.get("DIRECT"):       # DIRECT=<shadingSamples>: the `direct` integrator on the same scene
        from mitsuba_amd.integrator import DirectHIP
        integ = DirectHIP(shadingSamples=int(os.environ["DIRECT"]))
    else:
        integ = PathHIP(maxDepth=md)
    film = HDRFilm(w, h)
    skw = {}
    if os.environ.get("SAMPLER") in ("sobol", "halton", "hammersley"):       # the reference's QMC samplers (tables: tests/golden, as the tests pass them)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import conftest as CT
        skw = dict(sobol=CT.sobol_tables(w, h)) if os.environ["SAMPLER"] == "sobol" else dict(sampler=getattr(A, "PHIP_SAMPLER_" + os.environ["SAMPLER"].upper()), qmc=CT.qmc_tables(-1))
    if not os.environ.get("NOWARM"):
        integ.render(sc, film, 1, flags=int(os.environ.get("FLAGS", "0"), 0), **skw)
    for _ in range(int(os.environ.get("REPEAT", 1)) - 1):
        integ.render(sc, film, spp, flags=int(os.environ.get("FLAGS", "0"), 0), **skw)
    extra = int(os.environ.get("FLAGS", "0"), 0)       # e.g. FLAGS=0: PHIP_FLAG_NO_FUSED (the wavefront kernels on a scene of k_mega)
    t = time.time(); integ.render(sc, film, spp, flags=extra | (0 if os.environ.get('NOTIMING') else A.PHIP_FLAG_KERNEL_TIMING), **skw); dt = time.time() - t
    st = integ.stats.as_dict()
    n = w * h * spp
    print(json.dumps({"scene": key, "tris": sb.n_triangles, "accel": sc.accel_info().as_dict(), "scene_create_s": round(tb, 3), "spp": spp, "Msamples/s": round(n / 1e6 / dt, 1),
                      "Mrays/s": round((st["closest_rays"] + st["shadow_rays"]) / 1e6 / dt, 1), "mean_len": round(st["path_vertices"] / n, 2),
                      "nodes/closest": round(st["closest_node_visits"] / max(st["closest_rays"], 1), 1), "tris/closest": round(st["closest_triangle_tests"] / max(st["closest_rays"], 1), 1),
                      "nodes/shadow": round(st["shadow_node_visits"] / max(st["shadow_rays"], 1), 1),
                      "fused": st["fused"], "lib": os.environ.get("PHIP_LIB", ""),
                      "kernel_ms": {k: round(st[k], 1) for k in ("trace_kernel_ms", "shadow_kernel_ms", "shade_kernel_ms", "film_kernel_ms", "fused_kernel_ms")}, "wall_ms": round(dt * 1e3, 1),
                      "iters": st["iterations"], "trace_GBs_alg": round(st["trace_kernel_bytes"] / 1e9 / (max(st["trace_kernel_ms"], 1e-9) / 1e3), 1)}))
