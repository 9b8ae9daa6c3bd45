#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path_hip hot path on MI355X (BASELINE.json metric: Cornell box + Sponza class).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full pass of the hot path over the workload: render every camera sample of the frame
(generate -> trace -> shade -> shadow ... -> film) with the scene already resident in HBM.

N = 1 (the driver's BENCH run): `value` is BASELINE.json configs[1], the config the metric is quoted on -- Cornell box,
1024x1024, 256 spp, diffuse + area light, path maxDepth=-1 rrDepth=5, gaussian rfilter (K timed steps after W warm-up
steps) -- and the same line carries the other single-GPU configs under "workloads": C3 (Sponza-class atrium, 1920x1080,
64 spp, maxDepth 8), C4 (glass room, 1920x1080, 512 spp, maxDepth 16) and a 1/16-spp slice of C5, each timed the same way
(min(K, 3) steps after one warm-up step).

N > 1 (the driver's SCALE runs): `value` is the SAME config with N x the samples per pixel -- Cornell box, 1024x1024,
256 N spp -- so that every GPU renders as many samples as the one GPU of the N = 1 line (WEAK scaling: value(N) / value(1)
is N x the efficiency): the 32x32 blocks are dealt round-robin in the reference's spiral order over the ranks (one process
per GPU; north_star's "pixel blocks shard across the GPUs"), every rank renders its blocks into a private full-frame film,
and a single RCCL reduce(SUM) of the (R,G,B,alpha,weight) film onto rank 0 closes every step.  `value` = samples of the whole
job / max-over-ranks time.  The same line carries BASELINE.json configs[4] -- the atrium at 3840x2160, 1024 spp, ONE FIXED JOB
split N ways (strong scaling) -- under "workloads", beside the single-GPU rate of that job measured in the same run.
`--workload X` with N > 1 shards the fixed job X (strong scaling).
(Launched WITHOUT torchrun, `--gpus N` uses the library's own multi-device path instead: one host thread per GPU inside
phip_render_device and ncclReduce from C++ -- what the Mitsuba plugin uses.)

Prints ONE JSON line on rank 0 (see the driver contract).  `roofline` describes the dominant kernel of the `value`
workload with two honest fractions (neither can exceed 1): hbm = PMC-measured HBM bytes per launch / live launch time /
8 TB/s, valu = active lane-operations / lane-slots (SQ counters); `cpu_baseline` is the reference itself (oracle/_ref,
Mitsuba 0.6 compiled from /root/reference) -- or the CPU oracle port where that is not built -- timed on this node's host
cores on a bounded sample, BEFORE the GPU phase.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (scene builder name, width, height, spp, maxDepth)
    "cornell_1024x1024_256spp": ("cornell_box", 1024, 1024, 256, -1),                 # C2 (the metric's config)
    "cornell_256x256_16spp_md4": ("cornell_box", 256, 256, 16, 4),                    # C1
    "atrium_1920x1080_64spp_md8": ("atrium", 1920, 1080, 64, 8),                      # C3
    "glassroom_1920x1080_512spp_md16": ("glass_room", 1920, 1080, 512, 16),           # C4
    "atrium_3840x2160_1024spp_md8": ("atrium", 3840, 2160, 1024, 8),                  # C5: the multi-GPU job (strong scaling)
    "atrium_3840x2160_64spp_md8": ("atrium", 3840, 2160, 64, 8),                      # 1/16 of C5's samples per pixel: its single-GPU rate
    # the `direct` integrator on the same scenes
    "cornell_1024x1024_256spp_direct": ("cornell_box", 1024, 1024, 256, "direct:1"),
    # the small scene that is NOT the benchmark: the same box with a rough-copper and a glass block (all three leaf BSDF models; wavefront kernels)
    "cornell_mixed_1024x1024_256spp": ("cornell_mixed", 1024, 1024, 256, -1),
    "atrium_1920x1080_64spp_direct4": ("atrium", 1920, 1080, 64, "direct:4"),
    # round 6: the scenes between the LDS-resident boxes and the atrium -- the Cornell box with a glass and a rough-copper sphere (or two diffuse
    # spheres) of 1 k triangles: a tree of ~110 compressed 8-wide nodes that lives in L2; the fused kernel walks it from memory (k_mega, traversal form 4)
    "cornell_spheres_1k_1024x1024_64spp": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 24, "nlat": 12, "materials": True}),
    "cornell_spheres_1k_diffuse_1024x1024_64spp": ("cornell_spheres", 1024, 1024, 64, -1, {"nlon": 24, "nlat": 12, "materials": False}),
}
HEADLINE = "cornell_1024x1024_256spp"
EXTRA_SINGLE_GPU = ["atrium_1920x1080_64spp_md8", "glassroom_1920x1080_512spp_md16", "atrium_3840x2160_64spp_md8",
                    "cornell_mixed_1024x1024_256spp", "cornell_1024x1024_256spp_direct",
                    "cornell_spheres_1k_1024x1024_64spp", "cornell_spheres_1k_diffuse_1024x1024_64spp"]
MULTI_GPU_JOB = "atrium_3840x2160_1024spp_md8"
MULTI_GPU_SLICE = "atrium_3840x2160_64spp_md8"        # the same job at 1/16 of the samples per pixel: its single-GPU rate
MULTI_GPU_MAX_STEPS = 3
CPU_BASELINE_EXTRA = list(EXTRA_SINGLE_GPU)      # every workload of the line gets a bounded CPU figure of its own


def make_integrator(md):
    """md: maxDepth of the path tracer, or "direct:<shadingSamples>" """
    from mitsuba_amd.integrator import PathHIP, DirectHIP
    if isinstance(md, str):
        return DirectHIP(shadingSamples=int(md.split(":")[1])), "direct_hip shadingSamples=%s" % md.split(":")[1]
    return PathHIP(maxDepth=md), "path_hip maxDepth=%d rrDepth=5" % md


def oracle_params(md, spp):
    from mitsuba_amd import _abi as A
    if isinstance(md, str):
        n = int(md.split(":")[1])
        return A.default_render_params(spp=spp, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=n, bsdf_samples=n)
    return A.default_render_params(spp=spp, max_depth=md)


def build_desc(workload):
    from mitsuba_amd import _ffi, scene as S
    name, w, h, spp, md = WORKLOADS[workload][:5]
    sb = getattr(S, name)(w, h, _ffi.gaussian_filter(0.5), **(WORKLOADS[workload][5] if len(WORKLOADS[workload]) > 5 else {}))
    return sb.desc(), w, h, spp, md, sb.n_triangles


def profile_json(kind, workload):
    """newest committed profiles/*<kind>*<workload>*.json (written on the GPU box by tools/pmc_traffic.py / tools/pmc_valu.py), or None"""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*%s*.json" % (kind, workload))), reverse=True):
        try:
            return json.load(open(f)), os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


def cpu_baseline(workload, seconds_target=10.0, single_core_seconds=4.0):
    """The CPU baseline on a bounded sample of the same workload (same scene / film / integrator, reduced spp):
    the REFERENCE ITSELF when oracle/_ref is there (its own libcore + librender + plugins, compiled from /root/reference by
    oracle/Makefile.ref in the build container; the prebuilt files travel with the repository snapshot) -- its complete
    multi-threaded render, RenderJob -> BlockedRenderProcess on one LocalWorker per core; otherwise the oracle port."""
    desc, w, h, spp, md, _ = build_desc(workload)
    cores = os.cpu_count() or 1
    run = None
    try:
        from oracle import ref_ffi as R
        if not os.path.exists(R.LIB):
            raise RuntimeError("oracle/_ref is not built")
        rs = R.RefScene(desc)
        ref_run = lambda s: rs.render_job(oracle_params(md, s), threads=cores, want_image=False)[1]
        ref_run(1)                   # starts the Scheduler's workers, touches every plugin: not timed
        run, st = ref_run, None
        kind, what = "reference", ("Mitsuba 0.6 itself (oracle/_ref: the reference's sources compiled with g++ -O3 -march=x86-64-v3, IEEE "
                                   "float semantics; SAH kd-tree, `independent` sampler, 32x32 blocks, %d LocalWorkers)" % cores)
    except Exception as e:           # no prebuilt reference on this box: the restatement is the baseline
        print("bench.py: reference CPU baseline unavailable (%s); timing the oracle port" % e, file=sys.stderr)
    if run is None:
        from oracle import oracle_ffi as O
        osc = O.OracleScene(desc)
        last = {}

        def run(s):
            t = time.time()
            last["st"] = osc.render(oracle_params(md, s), threads=cores)[2]
            return time.time() - t
        kind, what = "port", "oracle = CPU restatement of the reference path: SAH kd-tree + Havran + TriAccel, %d threads" % cores
        st = last
    # calibrate on a short render (a 1-spp job is dominated by per-job set-up), then time ~seconds_target of work
    s0 = min(spp, 4)
    dt0 = max(run(s0), 1e-3)
    s = int(max(1, min(spp, round(seconds_target / (dt0 / s0)))))
    dt1 = run(s) if s != s0 else dt0
    out = {"value": round(w * h * s / 1e6 / dt1, 4), "unit": "Msamples/s", "cores": cores, "kind": kind,
           "sample": "%s at %d spp (%d samples, %.1f s; %s)" % (workload, s, w * h * s, dt1, what)}
    if st and st.get("st") is not None:
        out["mrays_per_s"] = round((st["st"].closest_rays + st["st"].shadow_rays) / 1e6 / dt1, 3)
    # the single-core rate beside it.  The reference's Scheduler keeps the workers its first job registered, so the one-worker job runs in a process of
    # its own (this file with --cpu-single-core): the same scene, camera and integrator on a film of 1/8 x 1/8 of the pixels
    if single_core_seconds > 0:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-single-core", workload, "--cpu-seconds", str(single_core_seconds)],
                               capture_output=True, text=True, timeout=120)
            out["single_core"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            out["single_core"] = {"value": None, "error": repr(e)}
    return out


def cpu_single_core(workload, seconds_target):
    """one worker of the reference (or of the oracle port) on the workload's scene at 1/8 x 1/8 of its film: prints one JSON object"""
    from mitsuba_amd import _ffi, scene as S
    name, w, h, spp, md = WORKLOADS[workload][:5]
    w8, h8 = max(32, w // 8), max(32, h // 8)
    desc = getattr(S, name)(w8, h8, _ffi.gaussian_filter(0.5), **(WORKLOADS[workload][5] if len(WORKLOADS[workload]) > 5 else {})).desc()
    try:
        from oracle import ref_ffi as R
        if not os.path.exists(R.LIB):
            raise RuntimeError("oracle/_ref is not built")
        rs = R.RefScene(desc)
        run = lambda s: rs.render_job(oracle_params(md, s), threads=1, want_image=False)[1]
        kind = "reference"
    except Exception:
        from oracle import oracle_ffi as O
        osc = O.OracleScene(desc)

        def run(s):
            t = time.time(); osc.render(oracle_params(md, s), threads=1); return time.time() - t
        kind = "port"
    dt0 = max(run(1), 1e-3)
    s = int(max(1, min(spp, round(seconds_target / dt0))))
    dt1 = run(s) if s != 1 else dt0
    print(json.dumps({"value": round(w8 * h8 * s / 1e6 / dt1, 5), "unit": "Msamples/s", "cores": 1, "kind": kind,
                      "sample": "%s on a %dx%d film at %d spp, one worker (%d samples, %.1f s)" % (workload, w8, h8, s, w8 * h8 * s, dt1)}))


def time_workload(workload, steps, warmup, rank, world, local, devices, D, torch):
    """Renders `workload` warmup + steps times on this rank's share (blocks rank, rank + world, ... in spiral order) and
    returns the per-rank aggregate.  Timed region: barrier + synchronize on both sides, max over ranks."""
    from mitsuba_amd import _abi as A
    from mitsuba_amd.integrator import Scene
    desc, W, H, spp, md, ntris = build_desc(workload)
    t_create = time.perf_counter()
    scene = Scene(desc, device=local)
    t_create = time.perf_counter() - t_create
    if devices:
        scene.replicate(devices)
    accel = scene.accel_info().as_dict()
    integ, integ_name = make_integrator(md)
    dev = torch.device("cuda", local)
    film = torch.zeros((H, W, 5), dtype=torch.float32, device=dev)
    extra = {"devices": devices} if devices else {}
    from mitsuba_amd.integrator import PinnedFilm
    host = PinnedFilm(W, H) if rank == 0 else None      # where the frame ends up: page-locked host memory (the metric includes the film's D2H)

    def step():
        """one frame, the film delivered to host memory on rank 0 -- the timed unit of `value`"""
        if world == 1:
            # phip_render: render (on `devices` inside the library when given) + the film's device-to-host copy into the pinned frame
            ok = integ.render_into(scene, host.ptr, spp, seed=0, flags=A.PHIP_FLAG_KERNEL_TIMING, **extra)
            assert ok
            return integ.stats
        ok = integ.render_device(scene, film.data_ptr(), spp, seed=0, shard_index=rank, shard_count=world, flags=A.PHIP_FLAG_KERNEL_TIMING)
        assert ok
        D.reduce_film(film, dst=0)
        if rank == 0:
            scene.film_to_host(film.data_ptr(), host.ptr)        # the merged frame to the pinned host frame (the master's film->put)
        return integ.stats

    def step_resident():
        """the same frame left in HBM (phip_render_device): `value_device_resident`"""
        ok = integ.render_device(scene, film.data_ptr(), spp, seed=0, shard_index=rank, shard_count=world, flags=A.PHIP_FLAG_KERNEL_TIMING, **extra)
        assert ok
        D.reduce_film(film, dst=0)
        return integ.stats

    for _ in range(warmup):
        step()
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg = {}
    step_s = []                                          # wall time of every timed step on this rank (a step returns with the frame in host memory: it is synchronous)
    for _ in range(steps):
        ts = time.perf_counter()
        st = step()
        step_s.append(time.perf_counter() - ts)
        for k, v in st.as_dict().items():
            agg[k] = agg.get(k, 0) + v
    D.barrier(); torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    total_samples = D.sum_over_ranks(agg["samples"], dev)
    total_rays = D.sum_over_ranks(agg["closest_rays"] + agg["shadow_rays"], dev)
    # beside it: the frame left in HBM, timed the same way on fewer steps
    rsteps = max(1, min(steps, 5))
    step_resident()
    D.barrier(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(rsteps):
        step_resident()
    D.barrier(); torch.cuda.synchronize()
    dt_res = D.max_over_ranks(time.perf_counter() - t1, dev) / rsteps
    scene.close()
    del film
    if host is not None:
        host.close()
    return {"workload": workload, "scene": WORKLOADS[workload][0], "triangles": ntris, "W": W, "H": H, "spp": spp, "integrator": integ_name,
            "accel": accel, "agg": agg, "dt": dt, "steps": steps, "warmup": warmup, "samples": total_samples, "rays": total_rays,
            "scene_create_ms": t_create * 1e3, "d2h_ms": agg.get("d2h_ms", 0.0) / max(steps, 1), "dt_resident_per_step": dt_res, "step_s": step_s}


def dominant_kernel(r):
    """(name, description, kernel ms over the timed steps, launches) of the kernel that takes most of the frame"""
    a = r["agg"]
    nodes = r["accel"]["n_nodes"]
    if a["fused"] and r["accel"].get("fused_traversal", 0) >= 4:
        return ("k_mega", "whole path in one persistent kernel, the compressed 8-wide tree (%d nodes of 80 B) and its 48-B Wald records walked from L2 through one shared task "
                "stack per wave in LDS; emitter table%s in LDS" % (nodes, " and materials" if r["accel"]["fused_traversal"] == 4 else ""), a["fused_kernel_ms"], max(int(a["iterations"]), 1))
    if a["fused"]:
        leaves = r["accel"]["n_leaves"]
        tree = ("the tree's %d leaves as a flat table of boxes (one uniform pass), " % leaves) if leaves <= 32 else ("BVH4 (%d nodes), " % nodes)
        return "k_mega", "whole path in one persistent kernel: " + tree + "Wald + shading records, emitters, materials in LDS", a["fused_kernel_ms"], max(int(a["iterations"]), 1)
    if a.get("vertex_traced"):
        return ("k_shade_trace", "one path vertex + its shadow ray + the next ray per slot and launch; the tree's %d leaves as a packed table of boxes, Wald records, "
                "emitters, materials in LDS, path state streamed through HBM once per vertex" % r["accel"]["n_leaves"], a["shade_kernel_ms"], max(int(a["iterations"]), 1))
    merged = a["shadow_kernel_ms"] == 0 and a["shadow_rays"] > 0        # closest-hit + any-hit rays in one persistent launch (big trees)
    wide = r["accel"]["node_bytes"] == 80                                # ... over the compressed 8-wide BVH
    rays = "k_rays_w" if wide else "k_rays_p"
    cands = {(rays if merged else ("k_trace_p" if nodes >= 64 else "k_trace")): a["trace_kernel_ms"], "k_shade": a["shade_kernel_ms"]}
    if not merged:
        cands["k_shadow_p"] = a["shadow_kernel_ms"]
    name = max(cands, key=cands.get)
    desc = {"k_rays_w": "closest-hit + any-hit traversal of the compressed 8-wide BVH, %d nodes of 80 B, 48-B Wald records" % nodes,
            "k_rays_p": "closest-hit + any-hit BVH4 traversal, %d nodes of 128 B, 48-B Wald records" % nodes,
            "k_trace_p": "closest-hit BVH4 traversal", "k_trace": "closest-hit BVH4 traversal", "k_shade": "path vertex shading over the pool",
            "k_shadow_p": "any-hit BVH4 traversal"}[name]
    return name, desc, cands[name], max(int(a["iterations"]), 1)


VMEM_LOADS_PER_NODE = {80: 5, 128: 7}     # 16-byte lane loads per node step: 80-byte wide node / 128-byte BVH4 node
VMEM_LOADS_PER_TRI = 3                      # ... per Wald record
VMEM_LOADS_PER_RAY = 2                      # ... per ray fetched (o | d)


def roofline(r):
    """Three measured fractions for the dominant kernel -- none can exceed 1 -- and `bound` = the largest:
       hbm:  HBM bytes per launch from the PMC pass committed under profiles/ (FETCH_SIZE + WRITE_SIZE, calibrated as the MI355X
             guide prescribes) / this run's average launch duration (HIP events on the library's stream) / 8 TB/s
       valu: active lane-operations / available lane-slots from the SQ counters committed under profiles/ (tools/pmc_valu.py)
       vmem: lane-level 16-byte load requests per second (counted by the kernel itself: node steps, triangle tests, rays fetched)
             / what the chip sustains when every lane gathers 16 bytes from an L1-resident set (tools/vmem_roof.py ->
             profiles/r03_vmem_roof.json); beside it the texture-data unit's busy share and the L1 / L2 hit rates of the kernel's
             own TA / TCP / TCC counters (tools/pmc_tcp.py -> profiles/r03_tcp_*.json)"""
    name, desc, kms, launches = dominant_kernel(r)
    a = r["agg"]
    nodes = r["accel"]["n_nodes"]
    avg_ms = kms / launches if launches else 0.0
    pw = r.get("profile_workload") or r["workload"]      # (the weak-scaling line of N GPUs: a rank's launch is the N = 1 config's launch -- the same number of samples through the same kernel)
    traffic, tsrc = profile_json("traffic", pw)
    valu, vsrc = profile_json("valu", pw)
    tcp, csrc = profile_json("tcp", pw)
    vroof, rsrc = profile_json("vmem_roof", "")
    def by_kernel(table):        # profile keys carry template arguments ("k_mega<0, false>"): match the kernel's base name
        for k, v in (table or {}).items():
            if k.split("<")[0] == name:
                return v
        return None
    # the PMC summaries are canned (counter passes cannot run inside the timed region): each carries the id of the library it was taken on
    from mitsuba_amd import _ffi as _F
    cur = _F.lib().phip_build_id().decode()
    ids = {"traffic": (traffic or {}).get("build_id"), "valu": (valu or {}).get("build_id"), "tcp": (tcp or {}).get("build_id")}
    stale = sorted(k for k, v in ids.items() if (traffic, valu, tcp)[("traffic", "valu", "tcp").index(k)] is not None and v != cur)
    tk = by_kernel((traffic or {}).get("kernels"))
    hbm_bytes = tk["hbm_bytes_per_launch"] if tk else None
    hbm_gbs = (hbm_bytes / 1e9) / (avg_ms / 1e3) if (hbm_bytes and avg_ms > 0) else None
    alg_per_launch = a["trace_kernel_bytes"] / launches if name in ("k_rays_w", "k_rays_p", "k_trace_p", "k_trace") else a["algorithmic_bytes"] / launches
    out = {"bound": None, "kernel": name + " (" + desc + ")",
           "achieved": round(hbm_gbs, 2) if hbm_gbs is not None else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(min(hbm_gbs / HBM_PEAK_GBS, 1.0), 5) if hbm_gbs is not None else None,
           "traffic": round(hbm_bytes, 1) if hbm_bytes else None, "traffic_source": tsrc,
           "avg_launch_ms": round(avg_ms, 5), "launches": launches,
           "algorithmic_bytes_per_launch": round(alg_per_launch, 1),
           "note": "achieved / frac = MEASURED HBM bytes per launch (PMC) / live launch time (/ 8 TB/s); the SURVEY 8(d) algorithmic bytes "
                   "(node + record fetches, ray / hit state) are listed beside it but are served by LDS / L1 / L2 / Infinity Cache, not by "
                   "the HBM pins.  `bound` names the largest of the three measured fractions hbm / valu / vmem",
           "kernel_ms_per_step": {k: round(a[k + "_kernel_ms"] / r["steps"], 3) for k in ("fused", "trace", "shadow", "shade", "film")},
           "stale_counters": bool(stale),
           "counters_build_ids": {"library": cur, **ids, "note": ("the PMC summaries under profiles/ named in *_source were taken on ANOTHER build than the one timed here (%s): "
                                                                  "launch times are live, bytes / lane counts are that build's" % ", ".join(stale)) if stale else
                                                                 "counter summaries and timed library are the same build"}}
    vk = by_kernel(valu)
    if vk:
        out["valu"] = {"frac": vk.get("valu_frac"), "issue_frac": vk.get("valu_issue_frac"), "lane_util": vk.get("lane_util"),
                       "wave_cycles_waiting": vk.get("wave_cycles_wait_frac"), "source": vsrc,
                       "definition": "frac = SQ_THREAD_CYCLES_VALU / (256 CU x 4 SIMD x 32 lanes x cycles); issue_frac = SQ_INSTS_VALU x 2 / (1024 x cycles)"}
    # vector-memory path: lane-level load requests against the measured roof
    peak = ((vroof or {}).get("peak_lane_random_16B") or {}).get("16KB_L1")
    if name in ("k_rays_w", "k_rays_p", "k_trace_p", "k_trace") and kms > 0 and (r["accel"]["node_bytes"] == 80 or nodes >= 64):      # (a tree staged in LDS makes no vector-memory requests for its nodes and records)
        per_node = VMEM_LOADS_PER_NODE.get(r["accel"]["node_bytes"], 7)
        loads = per_node * (a["closest_node_visits"] + a["shadow_node_visits"]) + VMEM_LOADS_PER_TRI * (a["closest_triangle_tests"] + a["shadow_triangle_tests"]) \
            + VMEM_LOADS_PER_RAY * (a["closest_rays"] + a["shadow_rays"])
    elif name in ("k_mega", "k_shade_trace"):
        loads = 0.0                  # nodes, records, tables and stack live in LDS; one 16-byte store per sample
    else:
        loads = None
    if loads is not None and peak and kms > 0:
        rate = loads / (kms / 1e3)
        ck = by_kernel((tcp or {}).get("kernels"))
        out["vmem"] = {"achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G lane-loads/s", "frac": round(min(rate / peak, 1.0), 4),
                       "lane_loads_per_ray": round(loads / max(a["closest_rays"] + a["shadow_rays"], 1), 2),
                       "peak_source": rsrc, "peak_definition": "every lane of the chip gathering 16 B from an L1-resident set, 8 waves per SIMD (L2-resident set: %.0f, Infinity-Cache-resident: %.0f G lane-loads/s)"
                       % (((vroof or {}).get("peak_lane_random_16B") or {}).get("2MB_L2", 0) / 1e9, ((vroof or {}).get("peak_lane_random_16B") or {}).get("16MB_MALL", 0) / 1e9)}
        if ck and ck.get("lines_per_instruction") and ck.get("clk_per_wave_instruction_per_cu") and ck.get("l1_hit_rate") is not None and ck.get("l2_hit_rate") is not None:
            # what the texture-data path allows for this kernel's line mix (the measured table of profiles/r03_vmem_roof.json): a 16-byte wave
            # instruction occupies the CU's return path for 17 clk, an L1 miss served by L2 costs 2.3 clk of the CU, one served by the Infinity Cache 7.6
            misses = ck["lines_per_instruction"] * (1.0 - ck["l1_hit_rate"])
            allowed = 17.0 + misses * (ck["l2_hit_rate"] * 2.3 + (1.0 - ck["l2_hit_rate"]) * 7.6)
            out["vmem"]["model_frac"] = round(min(allowed / ck["clk_per_wave_instruction_per_cu"], 1.0), 4)
            out["vmem"]["model_note"] = ("allowed / measured clk per wave instruction and CU = %.1f / %.1f for %.1f distinct lines per instruction at L1 / L2 hit rates %.2f / %.2f: "
                                         "the share of the texture-data path's throughput for THIS line mix (frac above is against fully divergent L1-resident gathers)"
                                         % (allowed, ck["clk_per_wave_instruction_per_cu"], ck["lines_per_instruction"], ck["l1_hit_rate"], ck["l2_hit_rate"]))
        if ck:
            out["vmem"].update({"td_busy_frac": ck.get("td_busy_frac"), "l1_hit_rate": ck.get("l1_hit_rate"), "l2_hit_rate": ck.get("l2_hit_rate"),
                                "lines_per_wave_instruction": ck.get("lines_per_instruction"), "clk_per_wave_instruction_per_cu": ck.get("clk_per_wave_instruction_per_cu"),
                                "counters_source": csrc})
    fr = {"hbm": out["frac"], "valu": (out.get("valu") or {}).get("frac"), "vmem": (out.get("vmem") or {}).get("frac")}
    fr = {k: v for k, v in fr.items() if v is not None}
    out["bound"] = max(fr, key=fr.get) if fr else "hbm"
    out["fractions"] = fr
    # ... and the SURVEY 8(d) figure as it is defined, beside the measured one: ALGORITHMIC bytes per launch / launch time / 8 TB/s.  It exceeds 1 where the bytes the
    # model counts (node and record fetches, ray / hit state) never reach the HBM pins; served_by names what delivers them instead
    if avg_ms > 0:
        out["frac_algorithmic"] = round(alg_per_launch / 1e9 / (avg_ms / 1e3) / HBM_PEAK_GBS, 4)
        wide_fused = name == "k_mega" and r["accel"].get("fused_traversal", 0) >= 4
        out["served_by"] = ("lds" if (name in ("k_mega", "k_shade_trace") and not wide_fused) else
                            "lds+l2" if wide_fused else
                            "l1+l2+mall" if name in ("k_rays_w", "k_rays_p", "k_trace_p", "k_trace", "k_shadow_p") else "hbm")
        out["frac_algorithmic_note"] = ("algorithmic_bytes_per_launch / avg_launch_ms / 8 TB/s (SURVEY 8(d)); > 1 means the byte model's traffic is served by `served_by`, "
                                        "not by HBM: the measured HBM fraction is `frac`")
    # the roof that binds this kernel, as a fraction of what that roof delivers: the texture-data path's model for the ray kernels (their line mix at their hit rates),
    # useful lane-operations / lane-slots for the kernels that live in registers and LDS, the measured HBM fraction for the streaming kernels
    if name in ("k_rays_w", "k_rays_p") and (out.get("vmem") or {}).get("model_frac") is not None:
        out["frac_bound"], out["bound_roof"] = out["vmem"]["model_frac"], "texture-data path (vmem.model_frac: allowed / measured clk per wave instruction for the kernel's line mix)"
    elif name in ("k_mega", "k_shade_trace") and (out.get("valu") or {}).get("frac") is not None:
        out["frac_bound"], out["bound_roof"] = out["valu"]["frac"], "VALU (valu.frac: active lane-operations / lane-slots; issue slots busy: valu.issue_frac)"
    elif out.get("frac") is not None:
        out["frac_bound"], out["bound_roof"] = max(fr.values()), "the largest measured fraction (%s)" % out["bound"]
    return out


def summary(r, world):
    a = r["agg"]
    msps = r["samples"] / 1e6 / r["dt"]
    ss = sorted(r.get("step_s") or [])
    spread = None
    if ss:
        n1 = r["samples"] / r["steps"] / 1e6
        med = ss[len(ss) // 2] if len(ss) % 2 else 0.5 * (ss[len(ss) // 2 - 1] + ss[len(ss) // 2])
        spread = {"ms_min": round(ss[0] * 1e3, 3), "ms_median": round(med * 1e3, 3), "ms_max": round(ss[-1] * 1e3, 3),
                  "value_best_step": round(n1 / ss[0], 3), "value_median_step": round(n1 / med, 3), "value_worst_step": round(n1 / ss[-1], 3),
                  "note": "wall time of the individual timed steps on rank 0 (a step returns with the frame in host memory); `value` is all steps / the bracketed time"}
    return {"value": round(msps, 3), "unit": "Msamples/s", "ms_per_step": round(r["dt"] / r["steps"] * 1e3, 3), "steps": r["steps"], "warmup": r["warmup"], "step_spread": spread,
            "mrays_per_s": round(r["rays"] / 1e6 / r["dt"], 1), "mean_path_length": round(a["path_vertices"] / max(a["samples"], 1), 3),
            "scene": r["scene"], "triangles": r["triangles"], "width": r["W"], "height": r["H"], "spp": r["spp"], "integrator": r["integrator"],
            "fused_kernel": bool(a["fused"]), "bvh_build_ms": round(r["accel"].get("build_ms", 0.0), 1), "scene_create_ms": round(r.get("scene_create_ms", 0.0), 1),
            "film_d2h_ms": round(r.get("d2h_ms", 0.0), 3),
            "value_device_resident": round(r["samples"] / r["steps"] / 1e6 / r["dt_resident_per_step"], 3) if r.get("dt_resident_per_step") else None,
            "value_note": "value = samples / wall time of phip_render INCLUDING the film's device-to-host copy into page-locked host memory (film_d2h_ms per frame, "
                          "SURVEY 8(d)); value_device_resident = the same frame left in HBM (phip_render_device)",
            "roofline": roofline(r)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS), help="time only this workload (default: the driver contract, see the module docstring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-single-core", default=None, choices=list(WORKLOADS), help="(helper process of cpu_baseline) time ONE CPU worker on this workload and print one JSON object")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--full-c5", action="store_true", help="N = 1: also time the whole 1024-spp 4K job (configs[4]) on one GPU, one step (~13 s): the same-build denominator of an N-GPU line")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the C3 / C4 / C5-slice workloads")
    args = ap.parse_args()
    if args.cpu_single_core:
        cpu_single_core(args.cpu_single_core, args.cpu_seconds)
        return

    import torch
    from mitsuba_amd import _ffi, distributed as D

    rank, world, local = D.init_from_env()
    in_library = world == 1 and args.gpus > 1          # no torchrun: the library's own multi-device path
    if world != args.gpus and not in_library:
        # a line whose n_gpus is not the --gpus that was asked for would be read as that N's measurement: refuse
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: launch with `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d`"
                         % (world, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: path_hip has no CPU fallback")
    if in_library and torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    n_gpus = args.gpus if in_library else world
    devices = list(range(args.gpus)) if in_library else None

    headline = args.workload or HEADLINE
    weak = n_gpus > 1 and not args.workload
    if weak:
        # the N-GPU line of the metric's config: N x the samples per pixel, the blocks dealt over N GPUs -- every GPU renders what the one GPU of the N = 1 line renders
        name, w_, h_, spp_, md_ = WORKLOADS[HEADLINE][:5]
        headline = "cornell_%dx%d_%dspp" % (w_, h_, spp_ * n_gpus)
        WORKLOADS[headline] = (name, w_, h_, spp_ * n_gpus, md_)
    cpu, cpu_extra = None, {}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(headline)                   # before the GPU phase: the GPU is busy for the rest of the run
        if not args.workload and not args.no_extra:
            for wl in CPU_BASELINE_EXTRA:
                cpu_extra[wl] = cpu_baseline(wl, seconds_target=8.0)

    steps, warmup = args.steps, args.warmup
    single = None
    main_r = time_workload(headline, steps, warmup, rank, world, local, devices, D, torch)
    if weak:
        main_r["profile_workload"] = HEADLINE
    extras = {}
    if n_gpus == 1 and not args.workload and not args.no_extra:
        for w in EXTRA_SINGLE_GPU + ([MULTI_GPU_JOB] if args.full_c5 else []):
            extras[w] = time_workload(w, 1 if w == MULTI_GPU_JOB else min(args.steps, 3), 1 if w != MULTI_GPU_JOB else 0, rank, world, local, devices, D, torch)
    if weak and not args.no_extra:
        # BASELINE.json configs[4]: the 4K atrium at 1024 spp as ONE FIXED JOB split N ways (strong scaling).  8.5 G samples per step (about 12 s on one GPU): a bounded
        # number of steps keeps every N inside the driver's time limit.  Its denominator: the SAME job on ONE GPU, unsharded, at 1/16 of the samples per pixel (the rate
        # does not depend on the sample count at this size: --full-c5 on the N = 1 line times all of it), measured in this very run by rank 0 alone
        if rank == 0:
            one = time_workload(MULTI_GPU_SLICE, 1, 1, 0, 1, local, None, D.Solo, torch)
            single = {"workload": MULTI_GPU_SLICE, "value": round(one["samples"] / 1e6 / one["dt"], 3), "unit": "Msamples/s", "ms_per_step": round(one["dt"] * 1e3, 3),
                      "note": "rank 0 alone, all blocks, before the timed region; same scene / film / integrator as the job, 64 of its 1024 samples per pixel"}
        D.barrier()
        extras[MULTI_GPU_JOB] = time_workload(MULTI_GPU_JOB, min(steps, MULTI_GPU_MAX_STEPS), min(warmup, 1), rank, world, local, devices, D, torch)

    if rank == 0:
        s = summary(main_r, world)
        out = {
            "metric": "Msamples/s", "value": s["value"], "unit": "Msamples/s", "n_gpus": n_gpus,
            "steps": steps, "warmup": warmup, "ms_per_step": s["ms_per_step"],
            # the default lines: every GPU renders the samples of the metric's config (N x its samples per pixel on N GPUs: weak scaling); `--workload X`: the fixed job X split N ways
            "higher_is_better": True, "scaling": "strong" if args.workload else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": headline, "scene": s["scene"], "scene_generator": "mitsuba_amd/scene.py rev 3 (round 3 shrank the atrium's balcony slabs by 3 cm: C3 / C4 / C5 numbers of rounds 1-2 are another scene)", "triangles": s["triangles"], "width": s["width"], "height": s["height"],
                       "spp": s["spp"], "integrator": s["integrator"], "rfilter": "gaussian stddev 0.5", "sampler": "ctr seed 0", "block_size": 32,
                       "parallelism": (("%s: 32x32 blocks dealt round-robin in spiral order over %d GPU(s), " % ("the metric's config at N x its samples per pixel" if weak else "one fixed job", n_gpus)) +
                                      ("one host thread per GPU inside libphip + ncclReduce(sum) of the film" if in_library else
                                       "one process per GPU + RCCL reduce(sum) of the film")) if n_gpus > 1 else "1 GPU"},
            "frame_ms": s["ms_per_step"], "step_spread": s["step_spread"], "mrays_per_s": s["mrays_per_s"], "mean_path_length": s["mean_path_length"],
            "fused_kernel": s["fused_kernel"], "roofline": s["roofline"],
            "build": {"library": _ffi.lib().phip_version().decode(), "id": _ffi.lib().phip_build_id().decode(),
                      "note": "id = hash of mitsuba_amd/csrc + include + compile flags, compiled into libphip.so and checked against the sources when it is loaded"},
        }
        if n_gpus > 1:
            out["requested"] = {"steps": args.steps, "warmup": args.warmup}
            out["roofline"]["scope"] = "the dominant kernel of ONE rank's share of the job (rank 0): the N-GPU line prices no collective -- the film reduce is reduce_ms of the library path / inside ms_per_step here"
            out["cpu_baseline_note"] = "no CPU baseline on the N > 1 line (the N = 1 line carries one per workload)"
            if weak:
                out["scaling_note"] = ("weak scaling: %d x the samples per pixel of the N = 1 line's config (%s) on %d GPUs -- value(N) / value(1) / N is the efficiency; "
                                       "the strong-scaling job of BASELINE.json configs[4] is workloads[%r], its one-GPU rate single_gpu_same_job there" % (n_gpus, HEADLINE, n_gpus, MULTI_GPU_JOB))
        if extras:
            out["workloads"] = {headline: {k: v for k, v in s.items() if k != "roofline"}}
            out["workloads"][headline]["roofline_frac_hbm"] = (s["roofline"] or {}).get("frac")
            for w, r in extras.items():
                out["workloads"][w] = summary(r, world)
                if w in cpu_extra:
                    out["workloads"][w]["cpu_baseline"] = cpu_extra[w]
            if weak and MULTI_GPU_JOB in extras:
                j = out["workloads"][MULTI_GPU_JOB]
                j["scaling"] = "strong"
                j["single_gpu_same_job"] = single
                sbal, ssrc = profile_json("shard_balance", MULTI_GPU_JOB)      # the shard balance taken on THIS job's shape on one GPU (round 5)
                pred = ((sbal or {}).get("N") or {}).get(str(n_gpus))
                if pred:
                    j["shard_balance"] = {"source": ssrc, "max_over_mean_shard_time": pred.get("max_over_mean"), "loss_to_imbalance": pred.get("loss_to_imbalance"),
                                          "reduce_s_estimate": pred.get("reduce_s_estimate"), "workload": (sbal or {}).get("workload"),
                                          "note": "tools/shard_balance.py: the N shards of the job rendered one after another on ONE GPU"}
        out["cpu_baseline"] = cpu
        print(json.dumps(out))


if __name__ == "__main__":
    main()
