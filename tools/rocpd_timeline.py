#!/usr/bin/env python3
"""GPU timeline analysis of a rocprofv3 kernel trace: busy/idle time, concurrency histogram, per-step makespan.
    python tools/rocpd_timeline.py x_results.db [n_last_kernels]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
# last third of the run = steady state
rows = rows[len(rows) * 2 // 3:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = []
for n, s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = {}
lvl, last = 0, t0
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - last)
    lvl += d; last = t
tot = float(t1 - t0)
print("window %.2f ms, %d kernels, sum of kernel time %.2f ms" % (tot / 1e6, len(rows), sum(e - s for _, s, e in rows) / 1e6))
for k in sorted(hist):
    print("  %d kernels in flight: %6.2f%% of wall" % (k, 100 * hist[k] / tot))
# time when ONLY kernel X runs (exclusive time) by kernel name
excl = {}
active = []
ev2 = []
for i, (n, s, e) in enumerate(rows):
    ev2.append((s, 0, i)); ev2.append((e, 1, i))
ev2.sort()
running = set(); last = t0
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:60]
for t, kind, i in ev2:
    if len(running) == 1:
        k = short(rows[next(iter(running))][0])
        excl[k] = excl.get(k, 0) + (t - last)
    if kind == 0: running.add(i)
    else: running.discard(i)
    last = t
print("exclusive (un-overlapped) time by kernel:")
for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-62s %7.2f ms (%.1f%%)" % (k, v / 1e6, 100 * v / tot))
