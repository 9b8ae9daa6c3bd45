#!/usr/bin/env python3
"""What does bit-exactness cost?    Times the product library against three variants built by tools/build_variant.sh
    fastc     -ffp-contract=fast                                                     (fused multiply-adds wherever the compiler finds them)
    native    -DPHIP_FMATH_NATIVE -fno-hip-fp32-correctly-rounded-divide-sqrt        (v_exp / v_log / v_sin / v_cos / v_rcp / v_rsq instead of the double-precision evaluations
                                                                                      of include/phip_fmath.h and the IEEE division / square root sequences)
    fastboth  both
on C2 (k_mega), the mixed Cornell box (k_mega<MM_ALL>), the spheres (k_mega on the wide tree) and C3 (k_shade + k_rays_w), and holds every variant's developed image against
Mitsuba 0.6 itself (tools/fullsize_vs_reference.py: C2, C3, the glass room at 960x540) -- north_star's bar is 1e-3 relative L2, not bit identity.
    python tools/price_exactness.py out.json          (on a GPU box; the variants must exist: mitsuba_amd/_build/libphip_<tag>.so)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("product", None), ("fastc", "libphip_fastc.so"), ("native", "libphip_native.so"), ("fastboth", "libphip_fastboth.so")]
SCENES = [("cornell", 256), ("cmixed", 256), ("sph1k", 64), ("atrium", 64)]
out = {"variants": {}}
for tag, lib in VARIANTS:
    env = dict(os.environ)
    if lib:
        path = os.path.join(ROOT, "mitsuba_amd", "_build", lib)
        if not os.path.exists(path):
            print("missing", path); continue
        env["PHIP_LIB"] = path
    v = out["variants"][tag] = {"rates": {}, "rel_l2_vs_mitsuba": {}}
    for scene, spp in SCENES:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_scenes.py"), scene], env=dict(env, SPP=str(spp), REPEAT="3"), capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            v["rates"][scene] = {"Msamples/s": d["Msamples/s"], "kernel_ms": d["kernel_ms"], "spp": spp}
        except Exception as e:
            v["rates"][scene] = {"error": repr(e), "stderr": r.stderr[-400:]}
        print(tag, scene, v["rates"][scene], flush=True)
    f = "/tmp/price_%s.json" % tag
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fullsize_vs_reference.py"), f, "C2 ", "C3", "C4-class"], env=env, capture_output=True, text=True)
    try:
        for k, x in json.load(open(f)).items():
            v["rel_l2_vs_mitsuba"][k] = {"rel_l2": x["rel_l2"], "pixels_differing_by_more_than_1e-3": x["pixels_differing_by_more_than_1e-3"], "gpu_seconds": x["gpu_seconds"]}
    except Exception as e:
        v["rel_l2_vs_mitsuba"] = {"error": repr(e), "stderr": r.stderr[-400:], "stdout": r.stdout[-400:]}
    print(tag, v["rel_l2_vs_mitsuba"], flush=True)
p = out["variants"].get("product", {}).get("rates", {})
for tag, v in out["variants"].items():
    v["speedup_vs_product"] = {s: round(v["rates"][s]["Msamples/s"] / p[s]["Msamples/s"], 3) for s in p if "Msamples/s" in v["rates"].get(s, {}) and "Msamples/s" in p[s]}
os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({t: {"speedup": v["speedup_vs_product"], "rel_l2": {k: x.get("rel_l2") for k, x in v["rel_l2_vs_mitsuba"].items() if isinstance(x, dict)}} for t, v in out["variants"].items()}, indent=1))
