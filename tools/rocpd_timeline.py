#!/usr/bin/env python3
"""GPU timeline analysis of a rocprofv3 kernel trace: busy/idle time, concurrency histogram, per-step makespan.
    python tools/rocpd_timeline.py x_results.db [n_last_kernels]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
has_q = "stream_id" in cols and "queue_id" in cols
rows_q = list(cur.execute("select %s, start, end%s from kernels order by start" % (name_col, ", stream_id, queue_id" if has_q else ", 0, 0")))
rows = [r[:3] for r in rows_q]
# last third of the run = steady state
rows_q = rows_q[len(rows_q) * 2 // 3:]
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
# idle gaps (no kernel in flight): total by (kernel that ended before, kernel that started after), and the gap-length histogram
gaps = {}
gh = {"<2us": 0, "2-5us": 0, "5-10us": 0, "10-20us": 0, ">20us": 0}
ends = sorted(rows_q, key=lambda r: r[1])
cur_end, prev_name, prev_sq = ends[0][2], ends[0][0], ends[0][3:]
for n, s, e, st, qu in ends[1:]:
    if s > cur_end:
        g = s - cur_end
        key = (short(prev_name), short(n))
        gaps[key] = gaps.get(key, (0, 0, 0, 0))
        gaps[key] = (gaps[key][0] + g, gaps[key][1] + 1, gaps[key][2] + int(prev_sq[0] == st), gaps[key][3] + int(prev_sq[1] == qu))
        b = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-20us" if g < 20000 else ">20us"
        gh[b] += g
    if e > cur_end:
        cur_end, prev_name, prev_sq = e, n, (st, qu)
print("idle time by gap length: " + ", ".join("%s %.2f ms" % (k, v / 1e6) for k, v in gh.items()))
print("largest idle gaps by (kernel before -> kernel after):")
for k, (v, c, ss, sq) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:16]:
    print("  %-44s -> %-44s %6.3f ms in %4d gaps (same stream %d, same hardware queue %d)" % (k[0][:44], k[1][:44], v / 1e6, c, ss, sq))
