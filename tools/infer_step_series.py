"""Round-6 probe: wall time of every bf16 inference forward of a bench-like loop, to find one-time stalls (python tools/infer_step_series.py [n])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch
from mask_cyclegan_vc.model import Generator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
torch.manual_seed(0)
gen = Generator().cuda()
xs = [torch.randn(16, 80, 512, device="cuda") for _ in range(4)]
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for i in range(n):
    h0 = time.perf_counter()
    ev[i][0].record()
    out = gen.infer(xs[i % 4], dtype="bf16")
    ev[i][1].record()
    host.append(1e3 * (time.perf_counter() - h0))
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t0)
gpu = [a.elapsed_time(b) for a, b in ev]
print("total wall %.1f ms for %d forwards (%.3f ms each); sum of per-forward GPU spans %.1f ms" % (wall, n, wall / n, sum(gpu)))
print("GPU span per forward (ms):", " ".join("%.2f" % g for g in gpu))
print("host enqueue per forward (ms):", " ".join("%.2f" % h for h in host))
