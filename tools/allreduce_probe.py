#!/usr/bin/env python3
"""All-reduce probe for the gradient exchange of the data-parallel step (RCCL over xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29561 tools/allreduce_probe.py

Measures, on the engine's own FlatGradReducer and communication stream:
  1. bus bandwidth of the bucket sizes the engine actually issues per iteration -- the generator's three parameter ranges
     (39 / 34 / 25 MiB per generator: up-sampling blocks + last conv, residual trunk, down-sampling head) and the 64 MiB buckets of the
     99 MB discriminator buffer -- as algbw = bytes / time and busbw = algbw * 2 (N - 1) / N (ring all-reduce);
  2. exposed time of one iteration's exchange (196.3 MB generator + 99.2 MB discriminator gradients) when it is issued behind a
     compute stream that is busy for `--compute-ms` (a spinning kernel stands in for the backward pass): serial (after the compute) vs
     overlapped (ranges released at 30 % / 60 % / 100 % of the compute, as the milestone events of mcvc_gen_backward_overlap do).

Expected on 8 x MI355X (7 xGMI links x ~153 GB/s per GPU, ring per-link bound): busbw ~ 250-320 GB/s on 25-64 MiB buckets, i.e. ~1.15 ms
for the generator exchange and ~0.6 ms for the discriminator's; with overlap the exposed part should be the tail
range only (~0.15-0.3 ms).  With MCVC_DIST_BACKEND=gloo (single-GPU dev box) the probe still runs and reports host-path numbers."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from mask_cyclegan_vc.parallel import FlatGradReducer, init_from_env  # noqa: E402

G_RANGES_MB = (39.3, 34.3, 24.6)          # per generator, in the order the last backward pass completes them
G_FLOATS = 49_075_458
D_FLOATS = 24_811_524


def timed(fn, iters, device):
    torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(device)
    dist.barrier()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--compute-ms", type=float, default=1.7, help="length of the stand-in for the last generator backward passes")
    ap.add_argument("--one-rank", action="store_true", help="single-GPU self-run: a ONE-rank nccl (= RCCL) group with the reducer forced on -- "
                    "proves library load (HSA_ENABLE_IPC_MODE_LEGACY=0), the communication stream and the event ordering; the 'bandwidth' of a "
                    "1-rank all-reduce is RCCL's launch + local copy cost, not xGMI")
    args = ap.parse_args()
    if args.one_rank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29573")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        rank, world, local_rank = 0, 1, 0
    else:
        rank, world, local_rank = init_from_env()
        if world < 2:
            raise SystemExit("run under torch.distributed.run with at least 2 ranks (or --one-rank on a single GPU)")
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    red = FlatGradReducer(force=args.one_rank)
    out = {"backend": dist.get_backend(), "world": world, "buckets": [], "iteration": {}}
    ring = 2.0 * (world - 1) / world
    for mb in sorted(set(G_RANGES_MB) | {64.0, 25.0, 34.6}):
        n = int(mb * (1 << 20) / 4)
        buf = torch.ones(n, device=dev)

        def one(buf=buf, n=n):
            red.reduce_range_after_(buf, 0, n, None)
            red.wait(dev)
        one()
        dt = timed(one, args.iters, dev)
        out["buckets"].append({"mib": mb, "ms": 1e3 * dt, "algbw_gbs": 4.0 * n / dt / 1e9, "busbw_gbs": 4.0 * n / dt / 1e9 * ring})
    # ---- one iteration's exchange behind a busy compute stream
    g = torch.ones(G_FLOATS, device=dev)
    d = torch.ones(D_FLOATS, device=dev)
    spin_cycles = int(args.compute_ms * 1e-3 * 100e6)          # torch.cuda._sleep counts ~100 MHz ticks on ROCm; calibrated below
    t0 = time.perf_counter(); torch.cuda._sleep(spin_cycles); torch.cuda.synchronize(dev)
    per_ms = spin_cycles / ((time.perf_counter() - t0) * 1e3)
    cyc = lambda ms: max(1, int(ms * per_ms))          # noqa: E731
    cuts = [0, int(0.40 * G_FLOATS), int(0.75 * G_FLOATS), G_FLOATS]          # 2 x (39 | 34 | 25) MB in completion order

    def serial():
        torch.cuda._sleep(cyc(args.compute_ms))
        red.reduce_(g)
        red.reduce_(d)

    def overlapped():
        evs = []
        for frac in (0.3, 0.3, 0.4):                      # milestones at 30 % / 60 % / 100 % of the compute
            torch.cuda._sleep(cyc(args.compute_ms * frac))
            e = torch.cuda.Event(); e.record(); evs.append(e)
        for k in range(3):
            red.reduce_range_after_(g, cuts[k], cuts[k + 1], evs[k])
        red.reduce_range_after_(d, 0, D_FLOATS, evs[2])
        red.wait(dev)

    def compute_only():
        torch.cuda._sleep(cyc(args.compute_ms))
    for name, fn in (("compute_only", compute_only), ("serial", serial), ("overlapped", overlapped)):
        fn()
        out["iteration"][name + "_ms"] = 1e3 * timed(fn, args.iters, dev)
    it = out["iteration"]
    it["exposed_serial_ms"] = it["serial_ms"] - it["compute_only_ms"]
    it["exposed_overlapped_ms"] = it["overlapped_ms"] - it["compute_only_ms"]
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
