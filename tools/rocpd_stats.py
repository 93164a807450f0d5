#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select %s, (end - start) from kernels" % name_col))
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print("# rocprofv3 --kernel-trace summary: %d dispatches, %.3f ms total GPU kernel time%s" % (
        len(rows), total / 1e6, (" (%d timed steps + warm-up + init)" % steps) if steps else ""))
    print("%-92s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-92s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (k, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / total))


if __name__ == "__main__":
    main()
