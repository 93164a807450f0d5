"""Round-6 probe (GPU box): is the InstanceNorm backward's counter-measured traffic (6x its launcher's count inside a bs=1 step,
profiles/r05_pmc_traffic_bs1.json) the kernel's own, or the write-back of its PRODUCER's dirty L2 lines falling into its dispatch window?

  python tools/norm_bwd_probe.py [alone|after_writer]     (plain: HIP-event durations;  under rocprofv3 --kernel-trace --pmc FETCH_SIZE /
                                                            WRITE_SIZE: tools/rocpd_pmc.py prints the per-dispatch averages)

`alone`: mcvc_instnorm_act_backward at the step's shapes, operands rotated through 12 buffer sets (cold in L2), nothing else in flight.
`after_writer`: every call is preceded by a kernel that leaves 24 MB of dirty lines in the L2s (a fill of a fresh buffer) -- what a
Winograd output transform / a K-split data gradient does in front of the norm backward inside the step.
Not part of the product path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
from mask_cyclegan_vc._hip import check, lib, ptr, stream  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "alone"
L = lib()
dev = torch.device("cuda:0")
# (label, N, C, H, W, act): act 1 = gated GLU (x has 2C channels), 2 = SiLU, 0 = none -- the bs=1 step's instances of norm_bwd_reg_kernel
CASES = [("<256,5> ds1 GLU 256x40x32 N=2", 2, 256, 40, 32, 1), ("<256,5> ds1 GLU N=1", 1, 256, 40, 32, 1),
         ("<256,20> up2 SiLU 128x80x64 N=2", 2, 128, 80, 64, 2), ("<64,5> ds2 GLU 256x20x16 N=2", 2, 256, 20, 16, 1),
         ("<64,2> D ds3 GLU 1024x10x8 N=2", 2, 1024, 10, 8, 1), ("<16,1> c1d2d 5120x1x16 N=2", 2, 5120, 1, 16, 0)]
SETS, REPS = 12, 24
for label, N, C, H, W, act in CASES:
    Cx = 2 * C if act == 1 else C
    bufs = []
    for _ in range(SETS):
        x = torch.randn(N, Cx, H, W, device=dev)
        m = x.mean(dim=(2, 3)); r = (x.var(dim=(2, 3), unbiased=False) + 1e-5).rsqrt()
        bufs.append(dict(x=x, stats=torch.stack([m, r], dim=2).contiguous(), dy=torch.randn(N, C, H, W, device=dev), dx=torch.empty_like(x)))
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    gg = torch.rand(C, device=dev) + 0.5 if act == 1 else None; bg = torch.randn(C, device=dev) if act == 1 else None
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    dgg = torch.zeros(C, device=dev) if act == 1 else None; dbg = torch.zeros(C, device=dev) if act == 1 else None
    dirty = [torch.empty(6 * 1024 * 1024, device=dev) for _ in range(4)]          # 24 MB each
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(REPS)]
    torch.cuda.synchronize()
    for i in range(REPS):
        s = bufs[i % SETS]
        if MODE == "after_writer":
            dirty[i % 4].fill_(float(i))
        ev[i][0].record()
        check(L.mcvc_instnorm_act_backward(ptr(s["x"]), ptr(g), ptr(b), ptr(gg), ptr(bg), ptr(s["stats"]), ptr(s["dy"]), ptr(s["dx"]),
                                           ptr(dg), ptr(db), ptr(dgg), ptr(dbg), N, C, H, W, act, stream()), "norm bwd")
        ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in ev[4:])
    el = N * C * H * W
    mb = 4.0 * el * ((4 if act == 1 else 2) + 1) / 1e6
    print("%-36s %-13s launcher bytes %7.3f MB   median %6.1f us  min %6.1f us  -> %6.0f GB/s" % (label, MODE, mb, t[len(t) // 2], t[0], mb / t[len(t) // 2] * 1e3))
