#!/usr/bin/env python3
"""Predicts the multi-GPU run on ONE GPU: renders the N shards of the C5 job's 64-spp slice (atrium 3840x2160, the
32x32 blocks dealt round-robin in the reference's spiral order) one after another for N = 2, 4, 8 and reports per-shard time, max / mean and

    predicted_scaling_efficiency(N) = T(1) / (N * (max_i T(i, N) + reduce(N)))

with reduce(N) = one ring reduce of the (H, W, 5) float32 film over xGMI at the per-link rate of the MI355X guide (7 links x ~153 GB/s per GPU; a ring
is per-link bound: 2 (N - 1) / N x bytes / 153 GB/s is the allreduce bound, a reduce onto one rank moves (N - 1) / N x bytes per link) -- a few
milliseconds against seconds of rendering, stated so that the measured number can be checked against it when a node appears.  What the prediction
cannot see: RCCL's launch overhead, host threads, clock differences between the eight chips.

    python tools/shard_balance.py [out.json]          (SPP=64 WORKLOAD=atrium4k in the environment)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device memory for the film)
from mitsuba_amd import _ffi, _abi as A, scene as S  # noqa: E402
from mitsuba_amd.integrator import Scene, PathHIP  # noqa: E402

W, H, spp, md = 3840, 2160, int(os.environ.get("SPP", 64)), 8
XGMI_LINK_GBS = 153.0
sb = S.atrium(W, H, _ffi.gaussian_filter(0.5))
sc = Scene(sb.desc()); integ = PathHIP(maxDepth=md)
film = torch.zeros((H, W, 5), dtype=torch.float32, device="cuda:0")


def render(i, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    assert integ.render_device(sc, film.data_ptr(), spp, shard_index=i, shard_count=n, flags=A.PHIP_FLAG_KERNEL_TIMING)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = integ.stats
    return {"s": dt, "samples": int(st.samples), "rays": int(st.closest_rays + st.shadow_rays), "iterations": int(st.iterations),
            "kernel_ms": round(st.trace_kernel_ms + st.shade_kernel_ms + st.film_kernel_ms, 2)}


render(0, 16)                                   # warm-up
t1 = min(render(0, 1)["s"] for _ in range(2))
out = {"workload": "atrium_3840x2160_%dspp_md8" % spp, "build_id": _ffi.lib().phip_build_id().decode(), "T1_s": round(t1, 4), "film_bytes": W * H * 20, "N": {}}
for n in (2, 4, 8):
    shards = [render(i, n) for i in range(n)]
    ts = [x["s"] for x in shards]
    reduce_s = (n - 1) / n * (W * H * 20) / (XGMI_LINK_GBS * 1e9)
    mx, mean = max(ts), sum(ts) / n
    out["N"][str(n)] = {"shard_s": [round(t, 4) for t in ts], "max_over_mean": round(mx / mean, 4), "sum_over_T1": round(sum(ts) / t1, 4),
                        "rays_max_over_mean": round(max(x["rays"] for x in shards) / (sum(x["rays"] for x in shards) / n), 4),
                        "iterations": [x["iterations"] for x in shards],
                        "reduce_s_estimate": round(reduce_s, 5),
                        "predicted_scaling_efficiency": round(t1 / (n * (mx + reduce_s)), 4),
                        "loss_to_imbalance": round(1 - mean / mx, 4), "loss_to_fixed_costs": round(1 - t1 / sum(ts), 4)}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
