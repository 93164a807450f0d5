#!/usr/bin/env python3
"""Per-launch trace of one generator inference forward (library's opt-in HIP-event trace): time, algorithmic GFLOP, TF/s, GB/s.
    python tools/infer_trace.py [bf16|f32] [B] [T]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch  # noqa: E402
from mask_cyclegan_vc import _hip  # noqa: E402
from mask_cyclegan_vc.model import Generator  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = int(sys.argv[3]) if len(sys.argv) > 3 else 512
torch.manual_seed(0)
g = Generator().cuda()
x = torch.randn(B, 80, T, device="cuda")
for _ in range(3):
    g.infer(x, None, dtype=dtype)
torch.cuda.synchronize()
L = _hip.lib()
buf = (ctypes.c_double * (4 * 4096))()
L.mcvc_trace_enable(1)
g.infer(x, None, dtype=dtype)
n = L.mcvc_trace_collect_raw(buf, 4096)
L.mcvc_trace_enable(0)
tot = 0.0
for i in range(n):
    k, ms, fl, by = buf[4 * i:4 * i + 4]
    tot += ms
    print("%3d %-22s %9.4f ms %10.3f GF %9.2f MB %8.1f TF/s %8.1f GB/s" % (
        i, L.mcvc_trace_kind_name(int(k)).decode(), ms, fl / 1e9, by / 1e6, fl / 1e9 / max(ms, 1e-6), by / 1e6 / max(ms, 1e-6)))
print("total %.3f ms in %d launches" % (tot, n))
