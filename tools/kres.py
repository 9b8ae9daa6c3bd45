#!/usr/bin/env python3
"""compact table of a -Rpass-analysis=kernel-resource-usage log:  python tools/kres.py log.txt [name filter]"""
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    print("%-44s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d occ %d" % (n, r.get("vgprs", -1), r.get("agprs", -1), r.get("sgprs", -1), r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
