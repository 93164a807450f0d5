#!/usr/bin/env python3
"""Weight-gradient microbenchmark through the C ABI with preallocated buffers (kernel + slab reduce only).

    python tools/wgrad_microbench.py [shape ...] [--iters 100]
Prints one line per (shape, batch): mean us over `iters` calls after 20 warm-up calls, TFLOP/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch  # noqa: E402
from mask_cyclegan_vc import _hip  # noqa: E402
from mask_cyclegan_vc._hip import lib, ptr, stream, check  # noqa: E402

SHAPES = {  # Cin, Cout(per wgrad call), KH, KW, stride, ph, pw, H, W
    "up2": (256, 512, 5, 5, 1, 2, 2, 40, 32), "up1": (256, 1024, 5, 5, 1, 2, 2, 20, 16),
    "ds1": (128, 256, 5, 5, 2, 2, 2, 80, 64), "ds2": (256, 256, 5, 5, 2, 2, 2, 40, 32),
    "conv1": (2, 128, 5, 15, 1, 2, 7, 80, 64), "last": (128, 1, 5, 15, 1, 2, 7, 80, 64),
    "d1": (1, 128, 3, 3, 1, 1, 1, 80, 64), "d2": (128, 256, 3, 3, 2, 1, 1, 80, 64), "d3": (256, 512, 3, 3, 2, 1, 1, 40, 32),
    "d4": (512, 1024, 3, 3, 2, 1, 1, 20, 16), "d5": (1024, 1024, 1, 5, 1, 0, 2, 10, 8), "dout": (1024, 1, 1, 3, 1, 0, 1, 10, 8),
}
ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*")
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--batches", default="1,2")
a = ap.parse_args()
L = lib()
for name in (a.shapes or list(SHAPES)):
    Cin, Cout, KH, KW, s, ph, pw, H, W = SHAPES[name]
    for N in [int(b) for b in a.batches.split(",")]:
        x = torch.randn(N, Cin, H, W, device="cuda")
        OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
        dy = torch.randn(N, Cout, OH, OW, device="cuda")
        dw = torch.zeros(Cout, Cin, KH, KW, device="cuda")
        n_slab = L.mcvc_conv2d_wgrad_slab_floats(N, Cin, H, W, Cout, KH, KW, s, ph, pw)
        slabs = torch.empty(max(n_slab, 1), device="cuda")
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(a.iters + 20):
            if i == 20:
                torch.cuda.synchronize(); ev0.record()
            check(L.mcvc_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(slabs), n_slab, N, Cin, H, W, Cout, KH, KW, s, ph, pw, stream()), "wgrad")
        ev1.record(); torch.cuda.synchronize()
        us = 1e3 * ev0.elapsed_time(ev1) / a.iters
        gf = 2.0 * N * OH * OW * Cout * Cin * KH * KW / 1e9
        print("%-6s N=%d  %8.1f us  %6.1f TF/s  (%.2f GF, slabs %.1f MB)" % (name, N, us, gf / us * 1e3, gf, n_slab * 4e-6))
