#!/usr/bin/env python3
"""The most frequent batched Winograd products of a bs=32 iteration, each alone on the chip.   MCVC_GEMM_CFG=<n> [GEMM_COLD=1] python tools/gemm_bs32_shapes.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
from mask_cyclegan_vc._hip import lib, ptr, stream
L = lib()
COLD = os.environ.get("GEMM_COLD", "0") != "0"
junk = torch.zeros(384 * 1024 * 1024, device="cuda") if COLD else None
SH = [("ds1 fwd/dgr", 36, 512, 2560, 512, 4), ("up2 fwd", 64, 512, 2560, 256, 3), ("up1 fwd", 64, 1024, 640, 256, 3), ("ds2 fwd", 36, 512, 640, 1024, 3),
      ("ds1 2B", 36, 512, 5120, 512, 2), ("up2 2B", 64, 512, 5120, 256, 1), ("up2 wgr", 64, 512, 256, 2560, 1), ("up2 dgr", 64, 256, 2560, 512, 1),
      ("up1 dgr", 64, 256, 640, 1024, 1), ("up1 wgr", 64, 1024, 256, 640, 1), ("ds1 wgr", 36, 512, 512, 2560, 1), ("ds2 dgr", 36, 1024, 640, 512, 1),
      ("ds2 wgr", 36, 512, 1024, 640, 1)]
tot = 0.0
for name, nxi, M, N, K, cnt in SH:
    a = torch.randn(nxi, K, M, device="cuda"); b = torch.randn(nxi, K, N, device="cuda"); c = torch.empty(nxi, M, N, device="cuda")
    call = lambda: L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), nxi, M, N, K, M, N, N, K * M, K * N, M * N, stream())   # noqa: E731
    rc = call()
    if rc:
        print(name, M, N, K, "rc", rc); continue
    for _ in range(2):
        call()
    ts = []
    for _ in range(6):
        if COLD:
            junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1))
    us = sorted(ts)[len(ts) // 2]
    gf = 2.0 * nxi * M * N * K / 1e9
    tot += us * cnt
    print("%-12s nxi=%3d M=%5d N=%5d K=%5d  %7.1f us  %6.1f TF/s  x%d" % (name, nxi, M, N, K, us, gf / us * 1e3, cnt), flush=True)
print("weighted total %.1f us" % tot)
