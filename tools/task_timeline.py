#!/usr/bin/env python3
"""Native (no profiler attached) per-task timeline of one training iteration: start / end of every task of the two phases' dependency
graphs on its lane, plus the optimizer steps on the caller's stream, from HIP events.

    python tools/task_timeline.py [--batch-size 1] [--steps 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch  # noqa: E402
from bench import build_nets, synthetic_batches  # noqa: E402
from mask_cyclegan_vc.engine import TrainEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-size", type=int, default=1)
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B = a.batch_size
eng = TrainEngine(build_nets(dev), B, 64)
bt = synthetic_batches(8, B, 64, 0, dev)
for i in range(8):
    eng.step(*bt[i % 8])
eng.flush()
torch.cuda.synchronize()
marks = []


def timed(name, fn):
    def run(*x, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*x, **k)
        e1.record()
        marks.append((name, e0, e1))
        return r
    return run


eng.generator_update = timed("adam G", eng.generator_update)
eng.discriminator_update = timed("adam D (+async pack)", eng.discriminator_update)
eng._timeline = []
t0 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(a.steps):
    eng.step(*bt[i % 8])
eng.flush()
t1 = torch.cuda.Event(enable_timing=True)
t1.record()
torch.cuda.synchronize()
print("# %d iterations, %.3f ms each (event pairs around every task add ~2 us each)" % (a.steps, t0.elapsed_time(t1) / a.steps))
rows = []
for ti, lane, rec, waits, e0, e1 in eng._timeline:
    rows.append((t0.elapsed_time(e0), t0.elapsed_time(e1), "lane %d" % lane, "task %2d rec=%s waits=%s" % (ti, rec, ",".join(waits))))
for name, e0, e1 in marks:
    rows.append((t0.elapsed_time(e0), t0.elapsed_time(e1), "main  ", name))
rows.sort()
for s, e, ln, what in rows:
    print("%8.3f -> %8.3f  (%6.3f ms)  %s  %s" % (s, e, e - s, ln, what))
