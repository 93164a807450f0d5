#!/usr/bin/env python3
"""Same-process A/B of an engine attribute (schedule choice) at one batch size:  python tools/ab_engine_attr.py 8 grouped_max_b 4 8"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
import bench
from mask_cyclegan_vc.engine import TrainEngine
from mask_cyclegan_vc.schedule import StepSchedule

B, attr, vals = int(sys.argv[1]), sys.argv[2], [int(v) for v in sys.argv[3:]]
dev = torch.device("cuda", 0)
batches = bench.synthetic_batches(16, B, 64, 0, dev)
steps = 30 if B <= 2 else (12 if B <= 8 else 6)
for rep in range(3):
    for v in vals:
        nets = bench.build_nets(dev)
        eng = TrainEngine(nets, B, 64, schedule=StepSchedule(generator_lr=2e-4, discriminator_lr=1e-4, num_epochs=6172, n_samples=81, batch_size=B, decay_after=2e5, stop_identity_after=1e4))
        setattr(eng, attr, v)
        eng._use(B)
        for i in range(4):
            eng.step(*batches[i % 16])
        eng.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.step(*batches[(4 + i) % 16]); eng.losses(lagged=2)
        eng.flush(); torch.cuda.synchronize()
        print("bs=%d %s=%d: %.3f ms" % (B, attr, v, 1e3 * (time.perf_counter() - t0) / steps), flush=True)
        del eng, nets
        torch.cuda.empty_cache()
