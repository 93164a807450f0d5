#!/usr/bin/env python3
"""Time the batched Winograd GEMM (mcvc_batched_gemm) on the shapes of a bs=1 step, per tile-width knob.
    python tools/gemm_tune.py            (MCVC_WINO_BN=0|32|64 is read at library load: one process per setting)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(36, 1024, 96, 256), (36, 256, 96, 1024), (36, 1024, 160, 256), (36, 256, 160, 1024), (16, 512, 96, 1024), (16, 1024, 96, 512),
          (16, 512, 160, 1024), (36, 512, 320, 256), (36, 256, 320, 512), (16, 512, 320, 512), (36, 512, 640, 256), (16, 512, 640, 512),
          (36, 1024, 256, 96), (36, 512, 256, 320)]
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
    import torch
    from mask_cyclegan_vc._hip import lib, ptr, stream
    L = lib()
    for nb, M, N, K in SHAPES:
        ldb = max(64, (N + 31) // 32 * 32)
        a = torch.randn(nb, K, M, device="cuda"); b = torch.randn(nb, K, ldb, device="cuda"); c = torch.empty(nb, M, ldb, device="cuda")
        call = lambda: L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), nb, M, N, K, M, ldb, ldb, K * M, K * ldb, M * ldb, stream())   # noqa: E731
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            call()
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 50
        gf = 2.0 * nb * M * N * K / 1e9
        print("%s nb=%2d M=%4d N=%3d K=%4d  %7.1f us  %6.1f TF/s" % (sys.argv[1], nb, M, N, K, us, gf / us * 1e3), flush=True)
else:
    for knob in sys.argv[1:] or ["0", "32", "64"]:
        pass
    res = {}
    for cfg in range(0, 12):
        env = dict(os.environ, MCVC_GEMM_CFG=str(cfg))
        r = subprocess.run([sys.executable, __file__, "cfg=%d" % cfg], env=env, capture_output=True, text=True, timeout=300)
        sys.stdout.write(r.stderr[-300:] if r.returncode else "")
        for ln in r.stdout.splitlines():
            key = ln.split("  ")[0].split(" ", 1)[1]
            res.setdefault(key, []).append((float(ln.split()[-4]), cfg))
    print("cfg: 0 default | 1 128x64 k16 s4 | 2 128x32 k16 s4 | 3 64x64 k16 s4 | 4 128x64 k32 s3 | 5 128x32 k32 s3 | 6 64x64 k32 s3 | "
          "7 128x64 k16 s6 | 8 128x32 k16 s6 | 9 64x64 k16 s6 | 10 128x64 k32 s4 | 11 64x64 k32 s4")
    for key, lst in res.items():
        print(key, " ".join("%d:%.1f" % (c, t) for t, c in lst), " best", min(lst))
