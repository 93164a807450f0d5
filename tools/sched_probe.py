"""Same-process timing of engine schedule variants at one batch size (r6): python tools/sched_probe.py B [steps]
  variants: default | grouped_max_b = B (grouped + pipelined at this batch) | grouped, phases back to back."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
from mask_cyclegan_vc.engine import TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402

B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
names = ("generator_A2B", "generator_B2A", "discriminator_A", "discriminator_B", "discriminator_A2", "discriminator_B2")


def batches(n):
    g = torch.Generator().manual_seed(1234)
    out = []
    for _ in range(n):
        b = []
        for _ in range(2):
            b.append(torch.randn(B, 80, 64, generator=g).cuda())
            m = torch.ones(B, 80, 64)
            for i in range(B):
                size = int(torch.randint(0, 25, (1,), generator=g)); start = int(torch.randint(0, 64 - size, (1,), generator=g))
                m[i, :, start:start + size] = 0
            b.append(m.cuda())
        out.append(b)
    return out


bt = batches(8)


def run(tag, **attrs):
    torch.manual_seed(0)
    nets = {n: (Generator() if i < 2 else Discriminator()).cuda() for i, n in enumerate(names)}
    eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=10 ** 6))
    for k, v in attrs.items():
        setattr(eng, k, v)
    eng._resid = None
    for i in range(5):
        eng.step(*bt[i % 8]); eng.losses(lagged=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(*bt[i % 8]); eng.losses(lagged=2)
    eng.flush()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print("bs=%d %-34s %.3f ms/step  (grouped %s pipelined %s) losses %s" % (B, tag, ms, eng._use_grouped(), eng._use_pipeline(),
                                                                               {k: round(v, 4) for k, v in eng.losses().items() if k in ("g_loss", "d_loss")}), flush=True)
    del eng, nets
    torch.cuda.empty_cache()


for rep in range(2):
    run("default")
    run("grouped_max_b=%d" % B, grouped_max_b=B)
    run("grouped, not pipelined", grouped_max_b=B, pipelined=False)
