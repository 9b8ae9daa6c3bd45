#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of libphip.so as the compiler reports it (hipcc -Rpass-analysis=kernel-resource-usage on the
units of mitsuba_amd/_ffi.py with their product flags; cross-compiles, no GPU) -> a markdown table (profiles/<round>_kernel_resources.md).
    python tools/kernel_resources.py > profiles/r03b_kernel_resources.md"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _ffi          # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out]


def waves_by_vgprs(v):
    return min(8, 512 // (-(-v // 8) * 8))


def blocks_by_sgprs(s):
    return min(8, 800 // (-(-s // 16) * 16 + 16))


print("# Kernel resources of libphip.so (build id %s; gfx950, `-Rpass-analysis=kernel-resource-usage`)\n" % _ffi.source_id())
print("waves/SIMD by VGPRs = min(8, 512 / VGPRs rounded up to 8); blocks of 256 per CU by SGPRs = min(8, 800 / (SGPRs rounded up to 16 + 16)) (MI355X guide).\n")
print("| kernel | VGPRs | SGPRs | scratch B/lane | static LDS B/block | waves/SIMD by VGPRs | blocks/CU by SGPRs |")
print("|---|---:|---:|---:|---:|---:|---:|")
flags = [f for f in _ffi.HIPCC_FLAGS if f != "-shared"]
for src, extra, obj in _ffi.UNITS:
    if obj not in ("phip.o", "phip_mega.o", "phip_megaw.o", "phip_megad.o", "phip_shade0_0.o", "phip_shade0_1.o", "phip_shade0_2.o", "phip_shade0_3.o"):       # the other shading units are the same kernels with more features compiled in
        continue
    r = subprocess.run([HIPCC] + flags + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(_ffi.CSRC, src), "-o", os.devnull],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}; rows.append(cur); continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = demangle([x["name"] for x in rows])
    for x, n in zip(rows, names):
        if not n.startswith("k_"):
            continue
        print("| `%s` | %d | %d | %d | %d | %d | %d |" % (n, x["vgprs"], x["sgprs"], x["scratch"], x["lds"], waves_by_vgprs(x["vgprs"]), blocks_by_sgprs(x["sgprs"])))
