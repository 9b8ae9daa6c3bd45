#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py dir [prefix]"""
import csv, collections, glob, sys, os
d = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(d, pre + "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    print("##", k)
    for c, x in sorted(v.items()):
        n = len(cnt[(k, c)])
        print("  %-28s total %.4g   per launch %.4g   (%d launches)" % (c, x, x / n, n))
