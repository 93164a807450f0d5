#!/usr/bin/env python3
"""Fixed cost vs per-k cost of the batched GEMM: one grid (nxi, M, N), K swept.   MCVC_GEMM_CFG=<n> python tools/gemm_k_sweep.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
from mask_cyclegan_vc._hip import lib, ptr, stream
L = lib()
for nxi, M, N in ((128, 256, 96), (32, 512, 320), (72, 1024, 96)):
    for K in (64, 128, 256, 512, 1024, 2048):
        a = torch.randn(nxi, K, M, device="cuda"); b = torch.randn(nxi, K, N, device="cuda"); c = torch.empty(nxi, M, N, device="cuda")
        call = lambda: L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), nxi, M, N, K, M, N, N, K * M, K * N, M * N, stream())   # noqa: E731
        if call():
            continue
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        print("nxi=%3d M=%5d N=%4d K=%5d  %7.1f us  %6.1f TF/s" % (nxi, M, N, K, us, 2.0 * nxi * M * N * K / us / 1e6), flush=True)
