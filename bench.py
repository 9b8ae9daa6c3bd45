#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path_hip hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full pass of the hot path over the workload: render every camera sample of the
frame (generate -> trace -> shade -> shadow ... -> film) with the scene already resident in HBM.
Workload (BASELINE.json configs[1], the config the metric is quoted on): Cornell box, 1024x1024,
256 spp, diffuse + area light, path maxDepth=-1 rrDepth=5, gaussian rfilter, synthetic scene.
With N GPUs the job is the same frame at 256*N spp whose 32x32 blocks are dealt round-robin (in
the reference's spiral order) to the N ranks -- per-GPU work is constant (weak scaling) -- and a
single RCCL reduce(SUM) of the (R,G,B,alpha,weight) film onto rank 0 closes every step.

Prints ONE JSON line on rank 0 (see the driver contract); `roofline` is for the dominant kernel
(k_trace, closest-hit BVH traversal) from HIP events recorded inside libphip on its own stream;
`cpu_baseline` is the CPU oracle ("port") timed on this node's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (scene builder name, width, height, spp, maxDepth)
    "cornell_1024x1024_256spp": ("cornell_box", 1024, 1024, 256, -1),
    "cornell_256x256_16spp_md4": ("cornell_box", 256, 256, 16, 4),
    "atrium_1920x1080_64spp_md8": ("atrium", 1920, 1080, 64, 8),
    "glassroom_1920x1080_512spp_md16": ("glass_room", 1920, 1080, 512, 16),
    # the `direct` integrator on the same scenes
    "cornell_1024x1024_256spp_direct": ("cornell_box", 1024, 1024, 256, "direct:1"),
    "atrium_1920x1080_64spp_direct4": ("atrium", 1920, 1080, 64, "direct:4"),
}


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


def build_desc(workload, spp_scale=1):
    from mitsuba_amd import _ffi, scene as S
    name, w, h, spp, md = WORKLOADS[workload]
    sb = getattr(S, name)(w, h, _ffi.gaussian_filter(0.5))
    return sb.desc(), w, h, spp * spp_scale, md, sb.n_triangles


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (tools/pmc_traffic.py), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*%s*.json" % workload)), reverse=True):
        try:
            k = json.load(open(f))["kernels"].get(kernel)
            if k:
                return round(k["hbm_bytes_per_launch"], 1)
        except Exception:
            pass
    return None


def cpu_baseline(workload, seconds_target=15.0):
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
    dt1 = max(run(1), 1e-3)
    s = int(max(1, min(spp, round(seconds_target / dt1))))
    if s > 1:
        dt1 = run(s)
    out = {"value": round(w * h * s / 1e6 / dt1, 4), "unit": "Msamples/s", "cores": cores, "kind": kind,
           "sample": "%s at %d spp (%d samples, %.1f s; %s)" % (workload, s, w * h * s, dt1, what)}
    if st and st.get("st") is not None:
        out["mrays_per_s"] = round((st["st"].closest_rays + st["st"].shadow_rays) / 1e6 / dt1, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1024x1024_256spp", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spp", type=int, default=0, help="override spp (debug only; makes the line non-comparable)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from mitsuba_amd import _ffi, _abi as A, distributed as D
    from mitsuba_amd.integrator import Scene, PathHIP

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("bench.py: WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d" % (world, args.gpus, args.gpus), file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: path_hip has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    desc, W, H, spp, md, ntris = build_desc(args.workload, spp_scale=world)
    if args.spp:
        spp = args.spp * world
    scene = Scene(desc, device=local)
    accel = scene.accel_info().as_dict()
    # the closest-hit kernel that runs for this scene (trees under 64 nodes use the per-slot launch)
    trace_kernel = "k_trace_p" if accel["n_nodes"] >= 64 else "k_trace"     # refined after the run: k_rays_p when the ray kernels are merged
    integ, integ_name = make_integrator(md)
    film = torch.zeros((H, W, 5), dtype=torch.float32, device=dev)
    flags = A.PHIP_FLAG_KERNEL_TIMING

    def step():
        ok = integ.render_device(scene, film.data_ptr(), spp, seed=0, shard_index=rank, shard_count=world, flags=flags)
        assert ok
        D.reduce_film(film, dst=0)
        return integ.stats

    for _ in range(args.warmup):
        step()
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg = {}
    for _ in range(args.steps):
        st = step()
        for k, v in st.as_dict().items():
            agg[k] = agg.get(k, 0) + v
    D.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    local_samples = agg["samples"]
    total_samples = D.sum_over_ranks(local_samples, dev)

    if rank == 0:
        msps = total_samples / 1e6 / dt
        ms_per_step = dt / args.steps * 1e3
        launches = max(int(agg["iterations"]), 1)
        trace_ms = agg["trace_kernel_ms"]
        merged = agg["shadow_kernel_ms"] == 0 and agg["shadow_rays"] > 0     # closest-hit + any-hit rays in one persistent launch (big trees)
        if merged:
            trace_kernel = "k_rays_p"
        kernel_desc = ("closest-hit + any-hit" if merged else "closest-hit") + " BVH4 traversal, %d nodes of 128 B, 48-B Wald records" % accel["n_nodes"]
        achieved = (agg["trace_kernel_bytes"] / 1e9) / (trace_ms / 1e3) if trace_ms > 0 else 0.0
        out = {
            "metric": "Msamples/s", "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": WORKLOADS[args.workload][0], "triangles": ntris, "width": W, "height": H,
                       "spp": spp, "spp_per_gpu_equivalent": spp // world, "integrator": integ_name,
                       "rfilter": "gaussian stddev 0.5", "sampler": "ctr seed 0", "block_size": 32,
                       "parallelism": "blocks round-robin over %d GPU(s) in spiral order + RCCL reduce(sum) of the film" % world},
            "frame_ms": round(ms_per_step, 3),
            "mrays_per_s": round((agg["closest_rays"] + agg["shadow_rays"]) / 1e6 / dt * world if world == 1 else
                                 D.sum_over_ranks(agg["closest_rays"] + agg["shadow_rays"], dev) / 1e6 / dt, 1),
            "mean_path_length": round(agg["path_vertices"] / max(agg["samples"], 1), 3),
            "roofline": {
                "bound": "hbm", "kernel": trace_kernel + " (" + kernel_desc + ")",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(args.workload, trace_kernel),
                "note": "achieved = ALGORITHMIC bytes (node + record fetches, ray in, hit out) / HIP-event kernel time; most of them are served by L1/L2/Infinity Cache, "
                        "traffic = PMC-measured HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, calibrated; profiles/)",
                "algorithmic_bytes_per_launch": round(agg["trace_kernel_bytes"] / launches, 1),
                "avg_launch_ms": round(trace_ms / launches, 5), "launches": launches,
                "whole_job_algorithmic_GBs": round(agg["algorithmic_bytes"] / 1e9 / dt, 2),
                "kernel_ms": {"trace": round(agg["trace_kernel_ms"], 2), "shadow": round(agg["shadow_kernel_ms"], 2),
                              "shade": round(agg["shade_kernel_ms"], 2), "film": round(agg["film_kernel_ms"], 2)},
            },
        }
    else:
        # keep collectives matched on the other ranks
        if world > 1:
            D.sum_over_ranks(agg["closest_rays"] + agg["shadow_rays"], dev)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))


if __name__ == "__main__":
    main()
