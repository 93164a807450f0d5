#!/usr/bin/env python3
"""Rate of the batched-GEMM kernel on the discriminators' strided-conv products at large batch (decides whether a staged-GEMM path pays).
    MCVC_GEMM_CFG=<n> python tools/gemm_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
from mask_cyclegan_vc._hip import lib, ptr, stream
L = lib()
SH = []
for NB in (8, 32, 64):
    SH += [("ds1 fwd", 512, NB * 1280, 1152), ("ds2 fwd", 1024, NB * 320, 2304), ("ds3 fwd", 2048, NB * 80, 4608),
           ("ds2 dgr", 2304, NB * 320, 1024), ("ds3 dgr", 4608, NB * 80, 2048),
           ("ds1 wgr", 512, 1152, NB * 1280), ("ds2 wgr", 1024, 2304, NB * 320), ("ds3 wgr", 2048, 4608, NB * 80)]
for name, M, N, K in SH:
    a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda"); c = torch.empty(M, N, device="cuda")
    call = lambda: L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), 1, M, N, K, M, N, N, K * M, K * N, M * N, stream())   # noqa: E731
    rc = call()
    if rc:
        print(name, M, N, K, "rc", rc); continue
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    gf = 2.0 * M * N * K / 1e9
    ref = a.t() @ b
    err = float((c - ref).norm() / ref.norm())
    print("%s M=%5d N=%6d K=%6d  %8.1f us  %6.1f TF/s  err %.1e" % (name, M, N, K, us, gf / us * 1e3, err), flush=True)
