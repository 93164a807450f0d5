#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd DB:  python tools/rocpd_pmc.py x_results.db [kernel substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
rows = list(cur.execute(q))
agg = {}
for k, c, v, n in rows:
    if pat in k:
        agg.setdefault(k[:70], {})[c] = (v / max(n, 1), n)
for k, d in agg.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %16.1f  (avg over %d dispatches)" % (c, v, n))
