#!/usr/bin/env python3
"""Fill the round-5 numbers block of DESIGN.md from the bench lines of one profile round.

    python tools/fill_design_r05.py profiles/r05          (prefix of bench_default.json, bench_bs{1,8,32}.json, bench_infer_bf16.json)

The block between the `r05-numbers` markers is regenerated from tools/design_r05_block.tmpl, so the script can be re-run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1]


def load(name):
    path = "%s%s.json" % (pre, name)
    with open(path) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


d = load("bench_default")
own = {b: load("bench_bs%d" % b) for b in (1, 8, 32)}
inf = load("bench_infer_bf16")
cfg = {c["config_id"].split(":")[0].split(" ")[0]: c for c in d["configs"]}
c2, c3, c4 = d["configs"][0], d["configs"][1], d["configs"][2]


def gb(x):
    return "%.2f" % (x / 1e9) if x else "n/a"


def bytes_cell(r):
    return "%s / %s / %s GB" % (gb(r.get("hbm_bytes_per_step_launcher")), gb(r.get("hbm_bytes_per_step_pmc")), gb(r.get("algorithmic_bytes_per_step")))


def roof(r):
    return "%.2f ms → %.2f; dominant family (%s) %.1f TF/s = %.3f of the fp32 MFMA peak" % (
        r["conv_roofline_ms"], r["frac_of_conv_roofline"], r["roofline"]["kernel"], r["roofline"]["achieved"], r["roofline"]["frac"])


def cpu(r):
    cb = r.get("cpu_baseline")
    if not cb:
        return "—"
    return "%.3g %s (%d threads) → %.0f×" % (cb["value"], cb["unit"].split(" ")[0], cb["cores"], r.get("speedup_vs_cpu", 0.0))


sc = d["schedule"]
post = d.get("after_identity_cutoff", {})
# per-family disagreement of the two byte counts at bs=32 (what DESIGN explains)
fam = own[32].get("hbm_bytes_per_step_by_family") or {}
worst = sorted(((k, v["pmc"] / max(v["launcher"], 1.0)) for k, v in fam.items() if v["launcher"] > 0), key=lambda kv: -kv[1])[:4]
note = ("Per family at bs=32 the counters read " + ", ".join("%s %.1f×" % kv for kv in worst) + " of the launchers' counts (bs=1: %.2f× in total, bs=32: %.2f×): "
        "the GEMM families gather an activation that every row-tile's workgroups re-read from beyond L2 (the MALL holds it: a 64-sample pass's "
        "largest activation is 94 MB), the InstanceNorm backward re-reads its conv output and slabs per channel group; the elementwise / Adam "
        "families, which stream, agree within 15 %%.  The counters' number is the one to hold against `algorithmic_bytes_per_step`: "
        "%.2f× at bs=1, %.2f× at bs=8, %.2f× at bs=32." % (
            own[1]["hbm_bytes_per_step_pmc"] / own[1]["hbm_bytes_per_step_launcher"], own[32]["hbm_bytes_per_step_pmc"] / own[32]["hbm_bytes_per_step_launcher"],
            own[1]["hbm_bytes_ratio_to_algorithmic"]["pmc"], own[8]["hbm_bytes_ratio_to_algorithmic"]["pmc"], own[32]["hbm_bytes_ratio_to_algorithmic"]["pmc"])
        if own[32].get("hbm_bytes_per_step_pmc") else "(no PMC summary for this revision.)")
rep = {
    "R5_C1_MS": "**%.2f**" % d["ms_per_step"], "R5_C1_ITS": "%.1f" % d["value"], "R5_C1_OWN": "%.2f" % own[1]["ms_per_step"], "R5_C1_ROOF": roof(d),
    "R5_C1_BYTES": bytes_cell(own[1]), "R5_C1_CPU": cpu(d),
    "R5_POST_MS": "%.2f" % post.get("ms_per_step", 0.0), "R5_POST_ITS": "%.1f" % post.get("iters_per_s", 0.0),
    "R5_SYNC_MS": "%.2f" % sc.get("sync_losses_ms_per_step", 0.0), "R5_SYNC_PCT": "%.1f" % (100.0 * sc.get("sync_losses_cost", 0.0)),
    "R5_C2_MS": "**%.1f**" % c2["ms_per_step"], "R5_C2_OWN": "%.1f" % own[32]["ms_per_step"], "R5_C2_ROOF": roof(c2), "R5_C2_BYTES": bytes_cell(own[32]), "R5_C2_CPU": cpu(c2),
    "R5_C3_MS": "**%.1f**" % c3["ms_per_step"], "R5_C3_OWN": "%.1f" % own[8]["ms_per_step"], "R5_C3_ROOF": roof(c3), "R5_C3_BYTES": bytes_cell(own[8]), "R5_C3_CPU": cpu(c3),
    "R5_C4_MS": "**%.2f** (%.2f M mel-frames/s)" % (c4["ms_per_step"], c4["value"] / 1e6), "R5_C4_OWN": "%.2f" % inf["ms_per_step"],
    "R5_C4_ROOF": "dominant conv %.0f TF/s = %.2f of the dense bf16 peak" % (c4["roofline"]["achieved"], c4["roofline"]["frac"]), "R5_C4_CPU": cpu(c4),
    "R5_PMC_NOTE": note,
}
block = open(os.path.join(ROOT, "tools", "design_r05_block.tmpl")).read()
for k in sorted(rep, key=len, reverse=True):
    block = block.replace(k, rep[k])
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
i, j = s.index("<!-- r05-numbers-begin -->\n") + len("<!-- r05-numbers-begin -->\n"), s.index("<!-- r05-numbers-end -->")
open(path, "w").write(s[:i] + block + s[j:])
# README: the round-5 table between its markers
readme = os.path.join(ROOT, "README.md")
r = open(readme).read()
table = (
    "<!-- r05-readme-begin -->\n"
    "| config (BASELINE.json) | default run / own process | roofline | note |\n|---|---|---|---|\n"
    "| C1 bs=1 full G+D iteration | **%.2f ms (%.1f it/s)** / %.2f; %d launches | step = %.2f of its convolution-FLOP floor at the fp32 MFMA peak (`conv_roofline_ms` %.2f); dominant family (%s) %.2f; HBM %s GB by the counters / %s by the launchers vs %s algorithmic | after the identity cut-off **%.2f ms (%.1f it/s)**; with the reference-exact loss readback %.2f ms; round 4 driver run 6.08; %.0fx the CPU oracle on the same box (32 threads) |\n"
    "| C2 bs=32 | **%.1f ms** / %.1f | %.2f of the conv floor; dominant family %.2f of the nominal peak, MFMA pipes 0.8 busy at the sustained clock | round 4 driver run 77.6 |\n"
    "| C3 per-GPU shape bs=8 | **%.1f ms** / %.1f | %.2f of the conv floor; dominant family %.2f (contains the small trunk products) | round 4 driver run 22.5 |\n"
    "| C4 generator inference bs=16 x 512 frames | bf16 **%.2f ms** (%.2f M mel-frames/s) / %.2f | bf16 convs %.2f of the dense peak at 2.4 GHz (the chip sustains 1.83-1.90 GHz under them) | round 4 driver run 3.72; bf16 vs fp32 oracle %.2e rel-L2 |\n"
    "<!-- r05-readme-end -->" % (
        d["ms_per_step"], d["value"], own[1]["ms_per_step"], d["kernel_launches_per_step"], d["frac_of_conv_roofline"], d["conv_roofline_ms"], d["roofline"]["kernel"],
        d["roofline"]["frac"], gb(own[1].get("hbm_bytes_per_step_pmc")), gb(own[1].get("hbm_bytes_per_step_launcher")), gb(own[1].get("algorithmic_bytes_per_step")),
        post.get("ms_per_step", 0.0), post.get("iters_per_s", 0.0), sc.get("sync_losses_ms_per_step", 0.0), d.get("speedup_vs_cpu", 0.0),
        c2["ms_per_step"], own[32]["ms_per_step"], c2["frac_of_conv_roofline"], c2["roofline"]["frac"],
        c3["ms_per_step"], own[8]["ms_per_step"], c3["frac_of_conv_roofline"], c3["roofline"]["frac"],
        c4["ms_per_step"], c4["value"] / 1e6, inf["ms_per_step"], c4["roofline"]["frac"], inf.get("parity_vs_cpu_rel_l2") or 0.0))
if "@@R05TABLE@@" in r:
    r = r.replace("@@R05TABLE@@", table)
else:
    i, j = r.index("<!-- r05-readme-begin -->"), r.index("<!-- r05-readme-end -->") + len("<!-- r05-readme-end -->")
    r = r[:i] + table + r[j:]
open(readme, "w").write(r)
print("filled from", pre)
