#!/usr/bin/env python3
"""HBM traffic per launch and per kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share
a pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python tools/pmc_traffic.py fetch_results.db write_results.db > profiles/r01_pmc_traffic.json

Corrections per the guide's HBM section: both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes, so it is doubled.  WRITE_SIZE is uncalibrated and used as reported.  Kernel families are the library's own
trace kinds (bench.py's `roofline.kernel`): conv_direct_kernel<2, 1, 2, 2, 5, false> -> conv_direct<2,1,2,2>."""
import json
import re
import sqlite3
import sys


def family(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").replace(" ", "")
    m = re.match(r"conv_direct_kernel<(\d+),(\d+),(\d+),(\d+),", n)
    if m:
        return "conv_direct<%s,%s,%s,%s>" % m.groups()
    m = re.match(r"conv_wgrad_kernel<(\d+),(\d+),", n)
    if m:
        return "conv_wgrad<%s,%s>" % m.groups()
    return re.sub(r"_kernel.*$", "", re.sub(r"\(.*$", "", n))


def per_family(dbfile, counter):
    db = sqlite3.connect(dbfile)
    q = "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? group by kernel_name, dispatch_id"
    agg = {}
    for k, _, v in db.execute(q, (counter,)):
        a = agg.setdefault(family(k), [0, 0.0])
        a[0] += 1; a[1] += v
    return agg


fetch = per_family(sys.argv[1], "FETCH_SIZE")
write = per_family(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(fam, (0, 0.0))
    nw, w = write.get(fam, (0, 0.0))
    rd = 2.0 * 1024.0 * f / max(nf, 1)
    wr = 1024.0 * w / max(nw, 1)
    out[fam] = {"launches_sampled": int(max(nf, nw)), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                "hbm_bytes_per_launch": rd + wr}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE on bench.py --serial (2 x FETCH_SIZE correction for gfx950)",
           "families": out}, sys.stdout, indent=1)
