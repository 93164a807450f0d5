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
    """Kernel name -> the library's trace kind (csrc/trace.cpp kNames: what bench.py's per-launch trace and `roofline.kernel` call it), so
    that the counters and the launchers' own byte counts are compared family by family."""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").replace(" ", "")
    m = re.match(r"conv_direct_kernel<(\d+),(\d+),(\d+),(\d+),", n)
    if m:
        return "conv_direct<%s,%s,%s,%s>" % m.groups()
    m = re.match(r"conv_wgrad_kernel<(\d+),(\d+),", n)
    if m:
        return "conv_wgrad<%s,%s>" % m.groups()
    base = re.sub(r"_kernel.*$", "", re.sub(r"\(.*$", "", n))
    base = re.sub(r"<.*$", "", base)
    table = (
        (("gemm2", "wino_gemm"), "wino_gemm"),
        (("sgemm", "igemm", "wgemm"), "sgemm"),
        (("update_net", "adam"), "adam"),
        (("pack_",), "pack"),
        (("norm_fwd", "bf16_norm", "bf16_stats", "bf16_finalize", "bf16_apply"), "norm_fwd"),
        (("norm_bwd",), "norm_bwd"),
        (("act_fwd", "disc_conv1_fwd", "disc_out_fwd"), "act_fwd"),
        (("act_bwd", "disc_out_dgrad"), "act_bwd"),
        (("trunk_",), "trunk_layer"),
        (("wgrad_smallk",), "wgrad_smallk"),
        (("wgrad_cin", "wgrad_cout"), "conv_wgrad<1,5>"),
        (("conv_fewout",), "conv_fewout"),
        (("bias_grad",), "bias_grad"),
        (("l1_loss", "lsgan_loss", "loss_combine"), "loss"),
    )
    for prefixes, kind in table:
        if any(base.startswith(p) for p in prefixes):
            return kind
    if base.startswith(("wino", "xform", "dw_accum", "mask_grad", "prep_input", "copy", "zero_words", "layout_conv", "draw_batch")):
        return "elementwise"
    return base


def dispatches(dbfile, counter):
    """[(dispatch_id, kernel name, counter value summed over its instances)] in dispatch order."""
    db = sqlite3.connect(dbfile)
    q = "select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id, kernel_name order by dispatch_id"
    return [(d, k, v) for d, k, v in db.execute(q, (counter,))]


def per_family(rows):
    agg = {}
    for _, k, v in rows:
        a = agg.setdefault(family(k), [0, 0.0])
        a[0] += 1; a[1] += v
    return agg


def short(name):
    return re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).strip()


def per_step(rows, nsteps, scale):
    """Bytes per training step by KERNEL INSTANCE (the full template instantiation, not the family average): every dispatch behind the
    last `pack_net_kernel` (the one-off re-pack of the six networks when the engine is built) belongs to one of the `nsteps` identical
    iterations the profiled command ran (warm-up + timed), so per instance  bytes per step = sum over its dispatches / nsteps."""
    last_pack = max([i for i, (_, k, _v) in enumerate(rows) if "pack_net_kernel" in k] or [-1])
    inst = {}
    for _, k, v in rows[last_pack + 1:]:
        a = inst.setdefault(short(k), [0, 0.0])
        a[0] += 1; a[1] += scale * v
    return {k: {"launches_per_step": n / nsteps, "bytes_per_launch": b / max(n, 1), "bytes_per_step": b / nsteps} for k, (n, b) in inst.items()}


frows, wrows = dispatches(sys.argv[1], "FETCH_SIZE"), dispatches(sys.argv[2], "WRITE_SIZE")
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fetch, write = per_family(frows), per_family(wrows)
out = {}
for fam in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(fam, (0, 0.0))
    nw, w = write.get(fam, (0, 0.0))
    rd = 2.0 * 1024.0 * f / max(nf, 1)
    wr = 1024.0 * w / max(nw, 1)
    out[fam] = {"launches_sampled": int(max(nf, nw)), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                "hbm_bytes_per_launch": rd + wr}
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE on bench.py --serial (2 x FETCH_SIZE correction for gfx950)",
       "families": out}
if nsteps > 0:
    rd, wr = per_step(frows, nsteps, 2.0 * 1024.0), per_step(wrows, nsteps, 1024.0)
    inst = {}
    for k in sorted(set(rd) | set(wr)):
        r, w = rd.get(k), wr.get(k)
        inst[k] = {"launches_per_step": (r or w)["launches_per_step"], "family": family(k),
                   "hbm_read_bytes_per_step": r["bytes_per_step"] if r else 0.0, "hbm_write_bytes_per_step": w["bytes_per_step"] if w else 0.0}
        inst[k]["hbm_bytes_per_step"] = inst[k]["hbm_read_bytes_per_step"] + inst[k]["hbm_write_bytes_per_step"]
    fam_step = {}
    for k, v in inst.items():
        fam_step[v["family"]] = fam_step.get(v["family"], 0.0) + v["hbm_bytes_per_step"]
    res["steps_profiled"] = nsteps
    res["instances"] = inst
    res["family_hbm_bytes_per_step"] = dict(sorted(fam_step.items(), key=lambda kv: -kv[1]))
    res["hbm_bytes_per_step_pmc"] = sum(v["hbm_bytes_per_step"] for v in inst.values())
json.dump(res, sys.stdout, indent=1)
