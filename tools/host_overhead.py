#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one step (no device sync inside the loop)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
from bench import build_nets, synthetic_batches
from mask_cyclegan_vc.engine import TrainEngine
dev = torch.device("cuda", 0)
eng = TrainEngine(build_nets(dev), 1, 64)
bt = synthetic_batches(4, 1, 64, 0, dev)
for mode in ("concurrent", "serial"):
    eng.concurrent = mode == "concurrent"
    for i in range(5): eng.step(*bt[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): eng.step(*bt[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: enqueue %.2f ms/step, total %.2f ms/step" % (mode, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))
