#!/usr/bin/env python3
"""Single-layer microbenchmark through the C ABI (for rocprofv3 --pmc runs).

    python tools/conv_microbench.py up2 --batch 2 --iters 20 [--op fwd|dgrad|wgrad]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch  # noqa: E402
from mask_cyclegan_vc import ops  # noqa: E402

SHAPES = {  # Cin, Cout, KH, KW, stride, ph, pw, H, W, shuffle
    "up2": (256, 512, 5, 5, 1, 2, 2, 40, 32, True), "up1": (256, 1024, 5, 5, 1, 2, 2, 20, 16, True),
    "ds1": (128, 512, 5, 5, 2, 2, 2, 80, 64, False), "ds2": (256, 512, 5, 5, 2, 2, 2, 40, 32, False),
    "dds3": (512, 1024, 3, 3, 2, 1, 1, 20, 16, False), "res": (256, 1024, 1, 3, 1, 0, 1, 1, 16, False),
    "last": (128, 1, 5, 15, 1, 2, 7, 80, 64, False), "conv1": (2, 256, 5, 15, 1, 2, 7, 80, 64, False),
}

ap = argparse.ArgumentParser()
ap.add_argument("shape")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--op", default="fwd")
a = ap.parse_args()
Cin, Cout, KH, KW, s, ph, pw, H, W, sh = SHAPES[a.shape]
N = a.batch
if a.shape == "res":
    H = N; N = 1
x = torch.randn(N, Cin, H, W, device="cuda")
w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.02
b = torch.randn(Cout, device="cuda")
OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
dy = torch.randn(N, Cout, OH, OW, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(a.iters + 3):
    if i == 3:
        torch.cuda.synchronize(); ev0.record()
    if a.op == "fwd":
        ops.conv2d_forward(x, w, b, s, (ph, pw), sh)
    elif a.op == "dgrad":
        ops.conv2d_dgrad(dy, w, tuple(x.shape), s, (ph, pw))
    else:
        ops.conv2d_wgrad(x, dy, tuple(w.shape), s, (ph, pw))
ev1.record(); torch.cuda.synchronize()
gf = 2.0 * N * OH * OW * Cout * Cin * KH * KW / 1e9
print("%s %s N=%d: %.1f us per call incl. pack/alloc (%.2f GF)" % (a.shape, a.op, a.batch, 1e3 * ev0.elapsed_time(ev1) / a.iters, gf))
