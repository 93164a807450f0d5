#!/usr/bin/env python3
"""Sweep the forward / data-gradient conv planner knobs per layer shape (one process; MCVC_CONV_TUNE=1 makes the library
re-read them on every call).  Time = conv kernels + the split-K slab reduce, from the library's own per-launch events.

    python tools/conv_tune.py [layer ...] [--batches 1,2] [--ops fwd,dgrad]
"""
import argparse
import ctypes
import itertools
import os
import sys

os.environ["MCVC_CONV_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch  # noqa: E402
from mask_cyclegan_vc._hip import lib, ptr, stream  # noqa: E402

LAYERS = {  # Cin, Cout(total, value|gate concatenated), KH, KW, stride, ph, pw, H, W, shuffle
    "g.conv1": (2, 256, 5, 15, 1, 2, 7, 80, 64, 0), "g.ds1": (128, 512, 5, 5, 2, 2, 2, 80, 64, 0), "g.ds2": (256, 512, 5, 5, 2, 2, 2, 40, 32, 0),
    "g.up1": (256, 1024, 5, 5, 1, 2, 2, 20, 16, 1), "g.up2": (256, 512, 5, 5, 1, 2, 2, 40, 32, 1), "g.last": (128, 1, 5, 15, 1, 2, 7, 80, 64, 0),
    "d.conv1": (1, 256, 3, 3, 1, 1, 1, 80, 64, 0), "d.ds1": (128, 512, 3, 3, 2, 1, 1, 80, 64, 0), "d.ds2": (256, 1024, 3, 3, 2, 1, 1, 40, 32, 0),
    "d.ds3": (512, 2048, 3, 3, 2, 1, 1, 20, 16, 0), "d.ds4": (1024, 2048, 1, 5, 1, 0, 2, 10, 8, 0), "d.out": (1024, 1, 1, 3, 1, 0, 1, 10, 8, 0),
    # the discriminator's real layers and the 1-D trunk as the network runs it at larger batch: ONE image of B rows x T/4 columns
    "D.ds1": (128, 256, 3, 3, 2, 1, 1, 80, 64, 0), "D.ds2": (256, 512, 3, 3, 2, 1, 1, 40, 32, 0), "D.ds3": (512, 1024, 3, 3, 2, 1, 1, 20, 16, 0),
    "t.vg8": (256, 1024, 1, 3, 1, 0, 1, 8, 16, 0), "t.vg16": (256, 1024, 1, 3, 1, 0, 1, 16, 16, 0), "t.vg64": (256, 1024, 1, 3, 1, 0, 1, 64, 16, 0),
    "t.out8": (512, 256, 1, 3, 1, 0, 1, 8, 16, 0), "t.out16": (512, 256, 1, 3, 1, 0, 1, 16, 16, 0), "t.out64": (512, 256, 1, 3, 1, 0, 1, 64, 16, 0),
    "t.c2d16": (5120, 256, 1, 1, 1, 0, 0, 16, 16, 0), "t.c1d16": (256, 5120, 1, 1, 1, 0, 0, 16, 16, 0),
}
ap = argparse.ArgumentParser()
ap.add_argument("layers", nargs="*")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batches", default="1,2")
ap.add_argument("--ops", default="fwd,dgrad")
ap.add_argument("--cfg", default="-1,0,1,2,3,4,5")
ap.add_argument("--nsplit", default="0,1,2,4,8,16")
ap.add_argument("--lds", default="39,52,78,156")
a = ap.parse_args()
L = lib()
L.mcvc_trace_kind_name.restype = ctypes.c_char_p
NK = L.mcvc_trace_kinds()
KINDS = [L.mcvc_trace_kind_name(k).decode() for k in range(NK)]
COUNT = [k for k, n in enumerate(KINDS) if n.startswith("conv_direct") or n in ("conv_fewout", "act_fwd")]
KNOBS = ("MCVC_CONV_CFG", "MCVC_CONV_NSPLIT", "MCVC_CONV_LDS_KB")
buf = (ctypes.c_double * (4 * NK))()


def timed(call, iters):
    for _ in range(3):
        if call():
            return None
    torch.cuda.synchronize()
    L.mcvc_trace_enable(1)
    for _ in range(iters):
        call()
    L.mcvc_trace_collect(buf)
    L.mcvc_trace_enable(0)
    return 1e3 * sum(buf[4 * k + 1] for k in COUNT) / iters


for name in (a.layers or list(LAYERS)):
    Cin, Cout, KH, KW, s, ph, pw, H, W, sh = LAYERS[name]
    for op in a.ops.split(","):
        if op == "dgrad" and name in ("g.conv1x",):
            continue
        for N in [int(b) for b in a.batches.split(",")]:
            OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
            x = torch.randn(N, Cin, H, W, device="cuda")
            w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.02
            b = torch.randn(Cout, device="cuda")
            dy = torch.randn(N, Cout, OH, OW, device="cuda")
            y = torch.empty(N, Cout, OH, OW, device="cuda")
            dx = torch.empty(N, Cin, H, W, device="cuda")
            wpack = torch.zeros(L.mcvc_conv2d_pack_floats(Cout, Cin, KH, KW), device="cuda")
            slabs = torch.empty(15 * max(y.numel(), dx.numel()), device="cuda")
            if op == "fwd":
                call = lambda: L.mcvc_conv2d_forward(ptr(x), ptr(w), ptr(b), ptr(y), ptr(wpack), ptr(slabs), 16, N, Cin, H, W, Cout, KH, KW,  # noqa: E731
                                                     s, ph, pw, sh, stream())
            else:
                call = lambda: L.mcvc_conv2d_dgrad(ptr(dy), ptr(w), ptr(dx), ptr(wpack), ptr(slabs), 16, N, Cin, H, W, Cout, KH, KW,  # noqa: E731
                                                   s, ph, pw, stream())
            gf = 2.0 * N * OH * OW * Cout * Cin * KH * KW / 1e9
            res = []
            for cfg, ns, lds in itertools.product(a.cfg.split(","), a.nsplit.split(","), a.lds.split(",")):
                for k, v in zip(KNOBS, (cfg, ns, lds)):
                    os.environ[k] = v
                t = timed(call, a.iters)
                if t is not None:
                    res.append((t, cfg, ns, lds))
            res.sort()
            base = [r for r in res if r[1:] == ("-1", "0", "78")]
            print("%-8s %-5s N=%d %.2f GF  default %.1f us | best:" % (name, op, N, gf, base[0][0] if base else float("nan")), flush=True)
            for t, cfg, ns, lds in res[:5]:
                print("      %7.1f us %5.1f TF/s  cfg=%s nsplit=%s lds=%s" % (t, gf / t * 1e3, cfg, ns, lds), flush=True)
