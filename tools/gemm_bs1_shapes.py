#!/usr/bin/env python3
"""The batched Winograd products of a bs=1 iteration, each alone on the chip (point count doubled = the two generators of a grouped launch).
    MCVC_GEMM_CFG=<n> python tools/gemm_bs1_shapes.py        (0 = the planner's choice)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
from mask_cyclegan_vc._hip import lib, ptr, stream
L = lib()
COLD = os.environ.get("GEMM_COLD", "0") != "0"
junk = torch.zeros(384 * 1024 * 1024, device="cuda") if COLD else None      # 1.5 GB: larger than L2 + MALL
SH = []
for ns in (1, 2):               # samples per pass
    t80, t320 = 96 if ns == 1 else 160, 320 * ns
    SH += [("ds1 fwd  %d" % ns, 16, 512, t320, 512), ("ds2 fwd  %d" % ns, 16, 512, t80, 1024), ("up1 fwd  %d" % ns, 36, 1024, t80, 256),
           ("up2 fwd  %d" % ns, 64, 512, t80, 256),
           ("up2 dgr  %d" % ns, 64, 256, t80, 512), ("up1 dgr  %d" % ns, 36, 256, t80, 1024), ("ds2 dgr  %d" % ns, 16, 1024, t80, 512),
           ("ds1 dgr  %d" % ns, 16, 512, t320, 512),
           ("up2 wgr  %d" % ns, 64, 512, 256, t80), ("up1 wgr  %d" % ns, 36, 1024, 256, t80), ("ds2 wgr  %d" % ns, 16, 512, 1024, t80),
           ("ds1 wgr  %d" % ns, 16, 512, 512, t320)]
tot = 0.0
for name, nxi, M, N, K in SH:
    nxi *= 2
    a = torch.randn(nxi, K, M, device="cuda"); b = torch.randn(nxi, K, N, device="cuda"); c = torch.empty(nxi, M, N, device="cuda")
    call = lambda: L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), nxi, M, N, K, M, N, N, K * M, K * N, M * N, stream())   # noqa: E731
    rc = call()
    if rc:
        print(name, M, N, K, "rc", rc); continue
    for _ in range(3):
        call()
    if COLD:                    # operands from HBM, as inside a training step (each weight set is read once per pass): flush L2 / MALL before every call
        ts = []
        for _ in range(8):
            junk.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1))
        us = sorted(ts)[len(ts) // 2]
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
    gf = 2.0 * nxi * M * N * K / 1e9
    ref = torch.bmm(a.transpose(1, 2), b)
    err = float((c - ref).norm() / ref.norm())
    tot += us
    print("%s nxi=%3d M=%5d N=%4d K=%5d  %7.1f us  %6.1f TF/s  err %.1e" % (name, nxi, M, N, K, us, gf / us * 1e3, err), flush=True)
print("total %.1f us" % tot)
