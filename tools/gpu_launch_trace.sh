#!/bin/bash
# per-launch durations of a workload's kernels in launch order (rocprofv3 --kernel-trace, csv): how the iterations of the wavefront loop are spent
#   bash tools/gpu_launch_trace.sh <gpu_scenes key> <spp> <out dir>
key=$1; spp=$2; o=$3; mkdir -p $o
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
NOWARM=1 SPP=$spp rocprofv3 --kernel-trace --output-format csv -d $o/prof -o trace -- python tools/gpu_scenes.py $key > $o/run.txt 2>&1
f=$(find $o/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $o/launches_$key.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for i, r in enumerate(rows):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print("%5d %-60s start %10.3f ms  dur %9.3f us" % (i, n, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
tail -2 $o/run.txt; rm -rf $o/prof
