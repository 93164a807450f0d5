#!/usr/bin/env python3
"""Small-N batched GEMM: time vs K (steady-state stage time vs fixed cost) and vs the number of batches (chip occupancy)."""
import sys
sys.path.insert(0, "maskcyclegan-vc_amd")
import torch
from mask_cyclegan_vc._hip import lib, ptr, stream
L = lib()
def run(nb, M, N, K):
    ldb = 96 if N <= 96 else 160
    a = torch.randn(nb * K * M, device="cuda"); b = torch.randn(nb * K * ldb, device="cuda"); c = torch.empty(nb * M * ldb, device="cuda")
    call = lambda: L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), nb, M, N, K, M, ldb, ldb, K * M, K * ldb, M * ldb, stream())
    rc = call()
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 50
    print("nb=%3d M=%4d N=%3d K=%4d rc=%d %7.1f us  %6.1f TF/s  A %.2f TB/s" % (nb, M, N, K, rc, us, 2.0 * nb * M * N * K / us / 1e6, 4.0 * nb * M * K / us / 1e6), flush=True)
for K in (64, 128, 256, 512, 1024, 2048):
    run(36, 1024, 96, K)
for nb in (4, 9, 18, 36, 72):
    run(nb, 1024, 96, 256)
