#!/usr/bin/env python3
"""Fill the @PLACEHOLDER@ numbers of DESIGN.md section 6 from the round's final bench lines (gpurun_out/<tag>/bench_*.json)."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02final"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", tag)
s = open(os.path.join(root, "DESIGN.md")).read()
for b in (1, 8, 32):
    r = json.load(open(os.path.join(d, "bench_bs%d.json" % b)))
    k = "BS%d" % b
    cb = r.get("cpu_baseline", {})
    s = s.replace("@%s_MS@" % k, "%.2f" % r["ms_per_step"]).replace("@%s_ITS@" % k, "%.1f" % r["value"])
    s = s.replace("@%s_FRAC@" % k, "%.2f" % r["step_mfma_fraction"]).replace("@%s_CPU@" % k, "%.2f" % cb.get("value", float("nan")))
    s = s.replace("@%s_X@" % k, "%.0f" % r.get("speedup_vs_cpu", float("nan"))).replace("@%s_LAUNCH@" % k, "%d" % r.get("kernel_launches_per_step", 0))
r = json.load(open(os.path.join(d, "bench_infer_bf16.json")))
s = s.replace("@INF_MS@", "%.2f" % r["ms_per_step"]).replace("@INF_FPS@", "%.0f" % (r["value"] / 1e3)).replace("@INF_FRAC@", "%.3f" % r["roofline"]["frac"])
open(os.path.join(root, "DESIGN.md"), "w").write(s)
print("filled")
