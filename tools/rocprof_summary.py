#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite database (--kernel-trace --stats) into a small markdown
table: per-kernel calls, total / average / min / max duration.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [title] > profiles/rNN_x.md
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc"
        % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# %s\n" % title)
    print("rocprofv3 --kernel-trace --stats (rocpd database `%s`), durations in microseconds\n" % sys.argv[1].split("/")[-1])
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, t, a, mn, mx in rows:
        short = n.split("(")[0].replace("void ", "")
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (short, c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
    print("\ntotal kernel time: %.2f ms" % (total / 1e6))


if __name__ == "__main__":
    main()
