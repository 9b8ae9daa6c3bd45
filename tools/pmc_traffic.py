#!/usr/bin/env python3
"""HBM traffic of the path_hip kernels from rocprofv3 PMC counters (run ON the GPU box).

    python tools/pmc_traffic.py <scene key of tools/gpu_scenes.py> <out.json> [spp] [bench workload name]

Runs (1) a calibration kernel pair with known HBM traffic and (2) one render of the workload, each under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as the MI355X guide
prescribes), and writes per-kernel bytes per launch with the calibration factors applied."""
import csv, glob, json, os, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, out = sys.argv[1], sys.argv[2]
spp = sys.argv[3] if len(sys.argv) > 3 else ""
bench_name = sys.argv[4] if len(sys.argv) > 4 else workload
tmp = os.path.join(ROOT, "gpurun_out", "pmc_traffic")
os.makedirs(tmp, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")

CAL_N, CAL_SRC = 1 << 26, 2 << 30        # 64M gathers from a 2 GiB buffer
cal_cmd = [sys.executable, "-c", "import ctypes,sys; sys.path.insert(0,%r); from mitsuba_amd import _ffi; L=_ffi.lib(); "
           "L.phip_debug_pmc_calibration.argtypes=[ctypes.c_size_t,ctypes.c_size_t]; assert L.phip_debug_pmc_calibration(%d,%d)==0" % (ROOT, CAL_SRC, CAL_N)]
ren_cmd = [sys.executable, os.path.join(ROOT, "tools", "gpu_scenes.py"), workload]       # (one render of the workload)
if spp:
    env["SPP"] = spp
env["NOWARM"] = "1"                     # no 1-spp warm-up render: a kernel with ONE launch per render (k_mega) would average it in


def run(counter, tag, cmd):
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", tag, "--output-format", "csv", "--"] + cmd,
                   check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(os.path.join(tmp, "**", tag + "_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k] += float(r["Counter_Value"]); n[k] += 1
    return agg, n


sys.path.insert(0, ROOT)
if os.environ.get("PHIP_POOL"):
    # the pool-size hook is read by experiment builds only (-DPHIP_EXPERIMENTS=1: expEnv is a constant in the product) -- a run against the product would
    # record an override that never applied
    _lib = os.environ.get("PHIP_LIB") or os.path.join(ROOT, "mitsuba_amd", "_build", "libphip.so")
    if not os.path.exists(_lib) or b"PHIP_POOL" not in open(_lib, "rb").read():
        sys.exit("pmc_traffic.py: PHIP_POOL is set, but %s does not read it (the product ignores algorithm-selecting environment variables): build an experiment "
                 "library with `tools/build_variant.sh exp -DPHIP_EXPERIMENTS=1` and pass it as PHIP_LIB, or unset PHIP_POOL" % _lib)
from mitsuba_amd import _ffi as _ffi_id  # noqa: E402  (the id compiled into the library that is being profiled: read from the file, no GPU call)
res = {"build_id": _ffi_id.built_id(os.environ.get("PHIP_LIB")), "workload": bench_name, "scene_key": workload, "spp_override": spp or None, "pool_slots_override": os.environ.get("PHIP_POOL"),
       "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB",
       "note": "per-launch figures: the path pool must have the size of the full job (8 M slots for jobs >= 256 M samples, else 4 M) -- set PHIP_POOL when spp is reduced"}
cf, _ = run("FETCH_SIZE", "cal_fetch", cal_cmd)
cw, _ = run("WRITE_SIZE", "cal_write", cal_cmd)
# gather: every 16-B read misses everything; the HBM transfer unit is a 64-B sector -> n*64 B expected (128 B if whole lines are fetched)
g_kib, s_kib = cf.get("k_debug_gather", 0), cf.get("k_debug_stream", 0)
res["calibration"] = {
    "gather_fetch_KiB": g_kib, "gather_expected_KiB_64B": CAL_N * 64 / 1024, "stream_fetch_KiB": s_kib, "stream_expected_KiB": CAL_N * 16 / 1024,
    "gather_write_KiB": cw.get("k_debug_gather", 0), "stream_write_KiB": cw.get("k_debug_stream", 0), "write_expected_KiB": CAL_N * 16 / 1024}
fetch_gather_corr = (CAL_N * 64 / 1024) / g_kib if g_kib else None
fetch_stream_corr = (CAL_N * 16 / 1024) / s_kib if s_kib else None
write_corr = (CAL_N * 16 / 1024) / cw["k_debug_stream"] if cw.get("k_debug_stream") else None
res["corrections"] = {"fetch_gather": fetch_gather_corr, "fetch_stream": fetch_stream_corr, "write": write_corr}
rf, nf = run("FETCH_SIZE", "ren_fetch", ren_cmd)
rw, nw = run("WRITE_SIZE", "ren_write", ren_cmd)
res["kernels"] = {}
for k in sorted(rf):
    if not k.startswith("k_"):
        continue
    gather = k.startswith(("k_trace", "k_shadow", "k_shade", "k_rays", "k_mega"))
    fc = (fetch_gather_corr if gather else fetch_stream_corr) or 1.0
    fb = rf[k] * 1024 * fc; wb = rw.get(k, 0) * 1024 * (write_corr or 1.0)
    res["kernels"][k] = {"launches": nf[k], "fetch_KiB_raw": rf[k], "write_KiB_raw": rw.get(k, 0), "fetch_correction": fc,
                         "hbm_bytes_per_launch": (fb + wb) / max(nf[k], 1)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["corrections"]), {k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in res["kernels"].items()})
