#!/usr/bin/env python3
"""Uninitialised-read hunt: poison the caching allocator's free blocks with a huge value, then run one generator phase
and report non-finite / changed results.   python tools/poison_check.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mcvc_oracle as orc  # noqa: E402
from mask_cyclegan_vc.engine import G_NAMES, D_NAMES, TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
val = float(sys.argv[2]) if len(sys.argv) > 2 else 1e30


def poison():
    xs = [torch.full((256 * 1024 * 1024,), val, device="cuda") for _ in range(24)]     # 24 GiB
    del xs


def run(poisoned):
    if poisoned:
        poison()
    nets = {}
    for i, n in enumerate(orc.NET_ORDER):
        m = Generator() if i < 2 else Discriminator()
        m.load_state_dict(orc.filler_params("G" if i < 2 else "D", 410 + i), strict=True)
        nets[n] = m.cuda()
    eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=64))
    rs = np.random.RandomState(77)
    batch = []
    for _ in range(2):
        batch += [torch.from_numpy(rs.randn(B, 80, 64).astype(np.float32)).cuda(), torch.ones(B, 80, 64, device="cuda")]
    out = []
    for which in ("G", "D"):
        for dst, src in zip(eng.static_in, batch):
            dst.copy_(src)
        eng._run_phase(which)
        lo = eng.losses()
        grp = eng.g_group if which == "G" else eng.d_group
        out.append((lo, grp.grad.double().clone().cpu()))
        if which == "G":
            eng.generator_update()
    return out


a = run(False)
b = run(True)
for (la, ga), (lb, gb), w in zip(a, b, "GD"):
    print(w, "losses", la, lb)
    print(w, "grad finite", bool(torch.isfinite(ga).all()), bool(torch.isfinite(gb).all()), "rel diff", float((ga - gb).norm() / ga.norm()))
