#!/usr/bin/env python3
"""Per-queue kernel timeline of ONE steady-state step from a rocprofv3 kernel-trace database (rocpd SQLite).
    python tools/rocpd_lanes.py x_results.db > lanes.txt
A step is delimited by consecutive occurrences of the first adam_kernel launch pair (two per step)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
sys.stderr.write("table %s cols %s\n" % (kt, cols))
name_col = "name" if "name" in cols else "kernel_name"
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
sel = "select %s, start, end%s from %s order by start" % (name_col, (", " + qcol) if qcol else "", kt)
rows = list(cur.execute(sel))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:44]


adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
# steady state: the step between the 3rd-last and the last pair of Adam launches
lo, hi = adam[-5] + 1, adam[-3] + 1
step = rows[lo:hi]
t0 = step[0][1]
queues = sorted({r[3] if qcol else 0 for r in step})
print("# one step: %d kernels, %.3f ms, queues %s" % (len(step), (max(r[2] for r in step) - t0) / 1e6, queues))
for r in step:
    q = queues.index(r[3]) if qcol else 0
    print("%9.1f %7.1f q%d %s%s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, q, "        " * q, short(r[0])))
