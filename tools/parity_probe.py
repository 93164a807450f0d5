"""Round-6 parity probe (GPU box): where does the HIP step stand against an fp64 anchor, next to the reference's own fp32 arithmetic?

  python tools/parity_probe.py ops      -> op-level rel-L2 of every scheme of the 5x5 layers against an fp64 convolution
  python tools/parity_probe.py cutoff   -> the `cutoff` fixture (4 iterations, bs=2): HIP (deterministic / default) and the fp32 oracle
                                           against the fp64 oracle, per tensor; JSON to gpurun_out/r06/parity_cutoff.json

Test infrastructure: imports oracle/ (never shipped, never timed)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "maskcyclegan-vc_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
OUT = os.path.join(ROOT, "gpurun_out", "r06")
os.makedirs(OUT, exist_ok=True)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm())


def ops():
    from mask_cyclegan_vc._hip import check, lib, ptr, stream
    L = lib()
    cases = [("up2", 256, 512, 1, 5, 5, 1, 2, 2, 1, 40, 32, True, (1, 2, 3)),
             ("up2.B2", 256, 512, 1, 5, 5, 1, 2, 2, 2, 40, 32, True, (1, 2, 3)),
             ("up1", 256, 1024, 1, 5, 5, 1, 2, 2, 1, 20, 16, True, (1, 2, 3)),
             ("up1.B2", 256, 1024, 1, 5, 5, 1, 2, 2, 2, 20, 16, True, (1, 2, 3)),
             ("ds1", 128, 256, 2, 5, 5, 2, 2, 2, 1, 80, 64, False, (1, 2, 3)),
             ("ds2", 256, 256, 2, 5, 5, 2, 2, 2, 1, 40, 32, False, (1, 2, 3)),
             ("ds2.B4", 256, 256, 2, 5, 5, 2, 2, 2, 4, 40, 32, False, (1, 2, 3)),
             ("d.ds2", 256, 512, 1, 3, 3, 2, 1, 1, 2, 40, 32, False, (3, 5))]
    rows = []
    for c in cases:
        name, Cin, Cout, nbr, KH, KW, s, ph, pw, N, H, W, shuffle, schemes = c
        g = torch.Generator().manual_seed(11)
        x = torch.randn(N, Cin, H, W, generator=g)
        ws = [torch.randn(Cout, Cin, KH, KW, generator=g) / np.sqrt(Cin * KH * KW) for _ in range(nbr)]
        bs = [torch.randn(Cout, generator=g) for _ in range(nbr)]
        wcat, bcat = torch.cat(ws, 0), torch.cat(bs, 0)
        ref64 = F.conv2d(x.double(), wcat.double(), bcat.double(), stride=s, padding=(ph, pw))
        ref32 = F.conv2d(x, wcat, bcat, stride=s, padding=(ph, pw))
        OH, OW = ref64.shape[2], ref64.shape[3]
        dy = torch.randn(N, nbr * Cout, OH, OW, generator=g)
        dx64 = torch.nn.grad.conv2d_input(x.shape, wcat.double(), dy.double(), stride=s, padding=(ph, pw))
        dx32 = torch.nn.grad.conv2d_input(x.shape, wcat, dy, stride=s, padding=(ph, pw))
        dw64 = torch.nn.grad.conv2d_weight(x.double(), wcat.shape, dy.double(), stride=s, padding=(ph, pw))
        dw32 = torch.nn.grad.conv2d_weight(x, wcat.shape, dy, stride=s, padding=(ph, pw))
        want64 = F.pixel_shuffle(ref64, 2) if shuffle else ref64
        want32 = F.pixel_shuffle(ref32, 2) if shuffle else ref32
        rows.append((name, "cpu-fp32", rel(want32, want64), rel(dx32, dx64), rel(dw32, dw64)))
        spec = (Cin, Cout, nbr, KH, KW, s, ph, pw)
        for scheme in schemes:
            packed = torch.zeros(L.mcvc_layer_packed_floats(*spec), device="cuda")
            scratch = torch.zeros(L.mcvc_layer_scratch_floats(N, H, W, *spec), device="cuda")
            wd, bd = [w.cuda() for w in ws], [b.cuda() for b in bs]
            w1, b1 = (wd[1], bd[1]) if nbr == 2 else (None, None)
            check(L.mcvc_layer_pack(ptr(wd[0]), ptr(bd[0]), ptr(w1), ptr(b1), ptr(packed), *spec, stream()), "layer_pack")
            xd, dyd = x.cuda(), dy.cuda()
            y = torch.full((N, nbr * Cout // 4, 2 * OH, 2 * OW) if shuffle else (N, nbr * Cout, OH, OW), float("nan"), device="cuda")
            rc = L.mcvc_layer_forward(ptr(xd), ptr(packed), ptr(wd[0]), ptr(w1), ptr(y), ptr(scratch), scratch.numel(), N, H, W, *spec, scheme,
                                      1 if shuffle else 0, stream())
            if rc != 0:
                rows.append((name, "scheme %d" % scheme, "refused rc=%d" % rc, "", ""))
                continue
            dx = torch.full((N, Cin, H, W), float("nan"), device="cuda")
            check(L.mcvc_layer_dgrad(ptr(dyd), ptr(packed), ptr(wd[0]), ptr(w1), ptr(dx), ptr(scratch), scratch.numel(), N, H, W, *spec, scheme,
                                     stream()), "dgrad")
            dws = [torch.zeros_like(w) for w in wd]
            check(L.mcvc_layer_wgrad(ptr(xd), ptr(dyd), ptr(dws[0]), ptr(dws[1]) if nbr == 2 else None, ptr(scratch), scratch.numel(), N, H, W,
                                     *spec, scheme, stream()), "wgrad")
            torch.cuda.synchronize()
            rows.append((name, "scheme %d" % scheme, rel(y, want64), rel(dx, dx64), rel(torch.cat([d.cpu() for d in dws], 0), dw64)))
    with open(os.path.join(OUT, "parity_ops.txt"), "w") as f:
        for r in rows:
            line = "%-8s %-10s fwd %s  dgrad %s  wgrad %s" % tuple(("%.3e" % v if isinstance(v, float) else str(v)) for v in r)
            print(line)
            f.write(line + "\n")


def cutoff(n_runs=2):
    import mcvc_oracle as orc
    from mask_cyclegan_vc import _hip
    from mask_cyclegan_vc.engine import TrainEngine
    from mask_cyclegan_vc.model import Discriminator, Generator
    from mask_cyclegan_vc.schedule import StepSchedule
    L = _hip.lib()
    gd = os.path.join(ROOT, "tests", "golden")
    js = json.load(open(os.path.join(gd, "step_cutoff.json")))
    bt = np.load(os.path.join(gd, "step_cutoff_batches.npz"))
    norms = json.load(open(os.path.join(gd, "grad_norms.json")))
    skip = {k.split(":", 1)[1] for k, v in norms.items() if v is not None and v < 1e-6}
    cfg = js["config"]
    bs, n_it = cfg["batch_size"], 4
    torch.set_num_threads(min(32, os.cpu_count() or 8))

    def oracle(dtype):
        onets = {n: orc.filler_params("G" if i < 2 else "D", s, dtype=dtype) for i, (n, s) in enumerate(zip(orc.NET_ORDER, cfg["filler_seeds"]))}
        so = orc.StepOracle(onets, skip_wasted=True)
        lam = [5, 5, 0, 0]
        losses = []
        for it in range(n_it):
            batch = [torch.from_numpy(bt["it%d_%s" % (it, k)]).to(dtype) for k in ("real_A", "mask_A", "real_B", "mask_B")]
            so.identity_lambda = float(lam[it])
            losses.append(so.step(*batch))
        return onets, losses

    cache = "/tmp/parity_probe_oracles.pt"
    if os.path.exists(cache):
        o64, l64, t64, o32, l32, t32 = torch.load(cache)
    else:
        t = time.time(); o64, l64 = oracle(torch.float64); t64 = time.time() - t
        t = time.time(); o32, l32 = oracle(torch.float32); t32 = time.time() - t
        torch.save((o64, l64, t64, o32, l32, t32), cache)
    print("fp64 oracle %.1f s, fp32 oracle %.1f s" % (t64, t32))

    def hip(det):
        was = L.mcvc_set_deterministic(1 if det else 0)
        try:
            nets = {}
            for i, (n, s) in enumerate(zip(orc.NET_ORDER, cfg["filler_seeds"])):
                m = Generator() if i < 2 else Discriminator()
                m.load_state_dict(orc.filler_params("G" if i < 2 else "D", s), strict=True)
                nets[n] = m.cuda()
            sched = StepSchedule(generator_lr=cfg["g_lr"], discriminator_lr=cfg["d_lr"], num_epochs=cfg["num_epochs"], n_samples=cfg["n_utt"],
                                 batch_size=bs, decay_after=cfg["decay_after"], stop_identity_after=cfg["stop_identity_after"])
            eng = TrainEngine(nets, bs, 64, schedule=sched)
            for it in range(n_it):
                eng.step(*[torch.from_numpy(bt["it%d_%s" % (it, k)]).cuda() for k in ("real_A", "mask_A", "real_B", "mask_B")])
            eng.flush()
            return {n: {k: p.detach().cpu().clone() for k, p in nets[n].named_parameters()} for n in nets}
        finally:
            L.mcvc_set_deterministic(was)

    runs = [("det%d" % i, hip(True)) for i in range(n_runs)] + [("fast%d" % i, hip(False)) for i in range(n_runs)]
    out = {"t64_s": t64, "t32_s": t32, "losses64": l64, "losses32": l32, "tensors": []}
    for name in orc.NET_ORDER:
        pnames = orc.generator_param_names() if name.startswith("gen") else orc.discriminator_param_names()
        for j, pn in enumerate(pnames):
            if pn in skip or pn.startswith("downSample4"):
                continue
            ref64 = o64[name][pn]
            if ref64.numel() == 1:
                continue
            row = {"net": name, "param": pn, "numel": ref64.numel(), "orc32_vs_64": rel(o32[name][pn], ref64)}
            for tag, r in runs:
                row[tag + "_vs_64"] = rel(r[name][pn], ref64)
                row[tag + "_vs_orc32"] = rel(r[name][pn], o32[name][pn])
            idx = torch.from_numpy(orc.sample_index(ref64.numel()))
            key = "final_%s_%d" % (name, j)
            if key in bt:
                fx = torch.from_numpy(bt[key].astype(np.float64))
                s64 = ref64.flatten()[idx]
                row["fixture_vs_64_samples"] = float((fx - s64).norm() / s64.norm())
                row["det0_vs_64_samples"] = float((runs[0][1][name][pn].flatten()[idx].double() - s64).norm() / s64.norm())
            out["tensors"].append(row)
    json.dump(out, open(os.path.join(OUT, "parity_cutoff%s.json" % os.environ.get("PROBE_TAG", "")), "w"), indent=1)
    T = out["tensors"]
    for tag in ["orc32"] + [r[0] for r in runs]:
        v = np.array([t[tag + "_vs_64"] for t in T])
        print("%-6s vs fp64: worst %.3e  median %.3e  n>1e-3 %d / %d" % (tag, v.max(), np.median(v), int((v > 1e-3).sum()), len(v)))
    ratio = np.array([t["det0_vs_64"] / max(t["orc32_vs_64"], 1e-12) for t in T])
    order = np.argsort(-ratio)
    print("largest HIP/ref error ratios (det0):")
    for i in order[:15]:
        t = T[i]
        print("  %-16s %-40s n=%8d  hip %.3e  ref %.3e  ratio %.2f" % (t["net"], t["param"], t["numel"], t["det0_vs_64"], t["orc32_vs_64"], ratio[i]))
    big = np.array([t["numel"] >= 4096 for t in T])
    print("ratio: median %.2f, 90%% %.2f, max %.2f; tensors >= 4096 elements: max %.2f" % (np.median(ratio), np.quantile(ratio, 0.9), ratio.max(), ratio[big].max()))
    d = np.array([rel_pair for rel_pair in [max(abs(t["det0_vs_64"] - t["det1_vs_64"]), 0) for t in T]]) if n_runs > 1 else None
    if d is not None:
        print("deterministic runs differ (should be 0):", float(d.max()))


if __name__ == "__main__":
    {"ops": ops, "cutoff": cutoff}[sys.argv[1]]()
