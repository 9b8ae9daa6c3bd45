#!/usr/bin/env python3
"""Summarises rocprofv3 --kernel-trace output into a small markdown table: per-kernel calls, total / average / min / max
duration.  Input: the rocpd SQLite database (default output format) or the *_kernel_trace.csv of --output-format csv.
    python tools/rocprof_summary.py <x_results.db | x_kernel_trace.csv> [title] > profiles/rNN_x.md
"""
import collections
import csv
import sys


def rows_from_db(path):
    import sqlite3
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    return cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc"
                       % (name_col, name_col)).fetchall()


def rows_from_csv(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = [(k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in d.items()]
    return sorted(rows, key=lambda r: -r[2])


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = rows_from_csv(path) if path.endswith(".csv") else rows_from_db(path)
    total = sum(r[2] for r in rows) or 1
    print("# %s\n" % title)
    print("rocprofv3 --kernel-trace (`%s`), durations in microseconds\n" % path.split("/")[-1])
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, t, a, mn, mx in rows:
        short = n.split("(")[0].replace("void ", "")
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (short, c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
    print("\ntotal kernel time: %.2f ms" % (total / 1e6))


if __name__ == "__main__":
    main()
