#!/usr/bin/env python3
"""Sweep the weight-gradient planner knobs per layer shape (one process; MCVC_WGRAD_TUNE=1 makes the library re-read
them on every call).  Prints, per (shape, batch), every configuration's time and the best one.

    python tools/wgrad_tune.py [shape ...] [--batches 1,2] [--iters 40]
"""
import argparse
import itertools
import os
import sys

os.environ["MCVC_WGRAD_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from mask_cyclegan_vc._hip import lib, ptr, stream  # noqa: E402

SHAPES = {  # Cin, Cout(per wgrad call), KH, KW, stride, ph, pw, H, W
    "up2": (256, 512, 5, 5, 1, 2, 2, 40, 32), "up1": (256, 1024, 5, 5, 1, 2, 2, 20, 16),
    "ds1": (128, 256, 5, 5, 2, 2, 2, 80, 64), "ds2": (256, 256, 5, 5, 2, 2, 2, 40, 32),
    "conv1": (2, 128, 5, 15, 1, 2, 7, 80, 64), "last": (128, 1, 5, 15, 1, 2, 7, 80, 64),
    "d1": (1, 128, 3, 3, 1, 1, 1, 80, 64), "d2": (128, 256, 3, 3, 2, 1, 1, 80, 64), "d3": (256, 512, 3, 3, 2, 1, 1, 40, 32),
    "d4": (512, 1024, 3, 3, 2, 1, 1, 20, 16), "d5": (1024, 1024, 1, 5, 1, 0, 2, 10, 8), "dout": (1024, 1, 1, 3, 1, 0, 1, 10, 8),
}
ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--batches", default="1,2")
ap.add_argument("--ms", default="0,1,2")
ap.add_argument("--waves", default="0,4,6,8,10,12,16")
ap.add_argument("--ksplit", default="0,1,2,3,4,6,8,16")
ap.add_argument("--lds", default="39,78,156")
a = ap.parse_args()
L = lib()
KNOBS = ("MCVC_WGRAD_MS", "MCVC_WGRAD_WAVES", "MCVC_WGRAD_KSPLIT", "MCVC_WGRAD_LDS_KB")


def timed(call, iters):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(iters + 5):
        if i == 5:
            ev0.record()
        rc = call()
        if rc:
            return None
    ev1.record(); torch.cuda.synchronize()
    return 1e3 * ev0.elapsed_time(ev1) / iters


for name in (a.shapes or list(SHAPES)):
    Cin, Cout, KH, KW, s, ph, pw, H, W = SHAPES[name]
    for N in [int(b) for b in a.batches.split(",")]:
        x = torch.randn(N, Cin, H, W, device="cuda")
        OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
        dy = torch.randn(N, Cout, OH, OW, device="cuda")
        dw = torch.zeros(Cout, Cin, KH, KW, device="cuda")
        slabs = torch.empty(64 * 1024 * 1024, device="cuda")      # 256 MB: never the limiter while tuning
        gf = 2.0 * N * OH * OW * Cout * Cin * KH * KW / 1e9
        res = []
        for ms, wv, ks, lds in itertools.product(a.ms.split(","), a.waves.split(","), a.ksplit.split(","), a.lds.split(",")):
            for k, v in zip(KNOBS, (ms, wv, ks, lds)):
                os.environ[k] = v
            t = timed(lambda: L.mcvc_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(slabs), slabs.numel(), N, Cin, H, W, Cout, KH, KW, s, ph, pw,
                                                  stream()), a.iters)
            if t is not None:
                res.append((t, ms, wv, ks, lds))
        res.sort()
        base = [r for r in res if r[1:] == ("0", "0", "0", "78")]
        print("%-6s N=%d %.2f GF  default %.1f us | best:" % (name, N, gf, base[0][0] if base else float("nan")), flush=True)
        for t, ms, wv, ks, lds in res[:6]:
            print("      %7.1f us %5.1f TF/s  ms=%s waves=%s ksplit=%s lds=%s" % (t, gf / t * 1e3, ms, wv, ks, lds), flush=True)
