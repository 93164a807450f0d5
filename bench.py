#!/usr/bin/env python3
"""bench.py -- MaskCycleGAN-VC full G+D training step on MI355X (BASELINE.json metric).

    python bench.py                                  # N=1, 50 timed steps after 10 warm-up
    python bench.py --batch-size 32                  # BASELINE configs[2]   (--batch-size 8: the per-GPU shape of configs[3])
    python bench.py --mode infer --dtype bf16        # BASELINE configs[4]: generator_A2B, bs=16, 80x512
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full iteration of the reference's inner loop (train.py:195-299: 10 generator forwards,
12 discriminator forwards, both backward passes, both Adam steps) on one synthetic minibatch per GPU
(bs=1, 80 mel x 64 frames, fp32 -- BASELINE.json configs[1]); inputs are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.  `value` is the whole-job rate: per-GPU bs=1 iterations
per second summed over the N data-parallel ranks (weak scaling).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
# multi-process GPU work on this image needs dmabuf IPC (RCCL / cross-process tensors fail with hipIpcGetMemHandle otherwise); exported on
# the boxes already -- set here, before the HIP runtime starts, in case a launcher's environment dropped it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALG_GFLOP_PER_SAMPLE_ITER = 504.1      # necessary conv MAC*2 work of one bs=1 iteration
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
# algorithmic bytes of one iteration (weights + conv / norm activation traffic of the necessary passes) + Adam's
# 2.07 GB (7 words x 73.9 M live parameters), at the tabulated batch sizes
ALG_BYTES_PER_ITER = {1: 4.9e9 + 2.07e9, 8: 18.4e9 + 2.07e9, 32: 64.7e9 + 2.07e9}
INFER_GFLOP_PER_SAMPLE_64 = 19.676     # generator forward, conv MAC*2, per sample of 64 frames


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def synthetic_batches(n_batches, B, T, rank, device, max_mask_len=25):
    """real ~ N(0,1) (the dataset is per-bin standardised); masks follow dataset/vc_dataset.py:51-55."""
    g = torch.Generator().manual_seed(1234 + rank)
    rs = np.random.RandomState(1234 + rank)
    out = []
    for _ in range(n_batches):
        ra = torch.randn(B, 80, T, generator=g)
        rb = torch.randn(B, 80, T, generator=g)
        ms = []
        for _k in range(2):
            m = np.ones((B, 80, T), dtype=np.float32)
            for b in range(B):
                size = rs.randint(0, max_mask_len)
                start = rs.randint(0, T - size)
                m[b, :, start:start + size] = 0.0
            ms.append(torch.from_numpy(m))
        out.append(tuple(t.to(device) for t in (ra, ms[0], rb, ms[1])))
    return out


def build_nets(device):
    from mask_cyclegan_vc.model import Discriminator, Generator
    torch.manual_seed(0)                   # reference construction order (train.py:103-110)
    names = ("generator_A2B", "generator_B2A", "discriminator_A", "discriminator_B", "discriminator_A2", "discriminator_B2")
    nets = {}
    for i, n in enumerate(names):
        nets[n] = (Generator() if i < 2 else Discriminator()).to(device)
    return nets


def pmc_traffic(kernel, B):
    """(HBM bytes per launch, source) of this kernel family from the committed rocprofv3 PMC passes (two separate --pmc passes,
    FETCH_SIZE x2 + WRITE_SIZE, summarised by tools/pmc_traffic.py into profiles/r<NN>_pmc_traffic_bs<B>.json together with the git
    revision they were taken at; the newest round's file wins).  The counters cannot be read from inside the bench, so this is a
    labelled constant, not a live measurement: `traffic_source` names file and revision; (None, None) when there is no summary."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic_bs%d.json" % B)), reverse=True):
        try:
            with open(path) as fh:
                js = json.load(fh)
            fam = js["families"].get(kernel)
            if fam is not None:
                return fam["hbm_bytes_per_launch"], "profiles/%s@%s" % (os.path.basename(path), js.get("git_rev", "?"))
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def pmc_bytes_per_step(B):
    """(HBM bytes per training step from the PMC passes, bytes per kernel family, source): the sum over kernel INSTANCES of the bytes of
    their dispatches in the profiled iterations (tools/pmc_traffic.py `hbm_bytes_per_step_pmc`; FETCH_SIZE x 2 + WRITE_SIZE, two separate
    passes).  Like `roofline.traffic` a labelled constant of the named revision -- the counters cannot be read inside the bench."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic_bs%d.json" % B)), reverse=True):
        try:
            with open(path) as fh:
                js = json.load(fh)
            if "hbm_bytes_per_step_pmc" in js:
                return js["hbm_bytes_per_step_pmc"], js.get("family_hbm_bytes_per_step"), "profiles/%s@%s" % (os.path.basename(path), js.get("git_rev", "?"))
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def trace_one_step(engine, batches):
    """Per-launch HIP-event trace of ONE iteration of the schedule that was timed, submitted in order on one stream (kernels do not
    overlap, so every launch's duration is its own).  With the pipelined step an iteration's worth of work is the discriminator phase of
    iteration t + the generator phase of iteration t+1: one untraced step leaves a pending discriminator phase, the next step is traced.
    Returns (rows per kernel family, raw per-launch list of (kernel, ms, flops, bytes))."""
    from mask_cyclegan_vc import _hip
    L = _hip.lib()
    cap = 16384
    buf = (ctypes.c_double * (4 * cap))()
    torch.cuda.synchronize()
    was = (engine._serial, engine.aux_wgrad)
    engine._serial, engine.aux_wgrad = True, False
    engine.step(*batches[0])
    torch.cuda.synchronize()
    L.mcvc_trace_enable(1)
    engine.step(*batches[1 % len(batches)])
    if engine._pending_D is None:              # (schedules without a pending phase: the step above was a whole iteration)
        pass
    n = L.mcvc_trace_collect_raw(buf, cap)
    L.mcvc_trace_enable(0)
    engine.flush()
    engine._serial, engine.aux_wgrad = was
    raw = [(L.mcvc_trace_kind_name(int(buf[4 * i])).decode(), buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n)]
    fam = {}
    for k, ms, fl, by in raw:
        r = fam.setdefault(k, {"kernel": k, "launches": 0, "ms": 0.0, "gflop": 0.0, "mbytes": 0.0})
        r["launches"] += 1; r["ms"] += ms; r["gflop"] += fl / 1e9; r["mbytes"] += by / 1e6
    rows = sorted(fam.values(), key=lambda r: -r["ms"])
    return rows, raw


def cpu_baseline(B, T, n_timed, first_losses, threads=0, warm=True):
    """The oracle (CPU restatement pinned to the reference, tests/test_oracle_golden.py) on the host cores.  ``warm=False`` (the extra
    configs of the default run, after the bs=1 baseline has warmed the process): no separate warm-up iteration -- the FIRST timed iteration
    doubles as the parity probe -- so that a bs=32 sample costs one iteration (~16 s), not two."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mcvc_oracle as orc
    # mkldnn oversubscribes badly on a 256-core host (one iteration did not finish in 10 minutes);
    # 32 threads is where the reference's CPU path runs fastest on these boxes
    torch.set_num_threads(threads if threads > 0 else min(os.cpu_count() or 1, 32))
    log("cpu baseline: %d threads" % torch.get_num_threads())
    nets = orc.default_init_nets(0)
    so = orc.StepOracle(nets)
    batches = synthetic_batches((1 if warm else 0) + n_timed, B, T, 0, "cpu")
    t0 = time.perf_counter()
    g0, d0 = so.step(*batches[0])                     # warm-up iteration (also the parity probe)
    t_first = time.perf_counter() - t0
    log("cpu %s iteration %.1f s" % ("warm-up" if warm else "first timed", t_first))
    t0 = time.perf_counter()
    for b in batches[1:]:
        so.step(*b)
    dt = time.perf_counter() - t0 + (0.0 if warm else t_first)
    parity = None
    if first_losses is not None:
        parity = {"g_loss_rel": abs(first_losses[0] - g0) / abs(g0), "d_loss_rel": abs(first_losses[1] - d0) / abs(d0)}
    return {"value": n_timed / dt, "unit": "iters/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d timed full G+D iterations at bs=%d 80x%d %s (%.1f s); oracle/mcvc_oracle.StepOracle, "
                      "reference autograd semantics incl. its discarded work"
                      % (n_timed, B, T, "after 1 warm-up" if warm else "in a process already warm from the bs=1 sample, first iteration", t_first),
            "host_cpus": os.cpu_count()}, parity


def infer_record(args, rank, world, device, dtype, B, T, steps, warmup):
    """BASELINE configs[4]: generator_A2B inference (test.py path), bs=16, 80 mel x 512 frames, all-ones mask, weights = the seeded
    default init cast to the compute dtype.  A step = one batched forward; value = mel-frames/s = 16*512 / latency, summed over ranks
    (independent replicas: inference has no exchange step).  Returns the record on rank 0 (None elsewhere)."""
    from mask_cyclegan_vc.model import Generator
    args = argparse.Namespace(**dict(vars(args), steps=steps, warmup=warmup))
    torch.manual_seed(0)
    gen = Generator().to(device)
    g = torch.Generator().manual_seed(1234 + rank)
    xs = [torch.randn(B, 80, T, generator=g).to(device) for _ in range(4)]
    log("infer: %s, bs=%d, %d frames" % (dtype, B, T))
    outs = None
    for i in range(args.warmup):
        outs = gen.infer(xs[i % len(xs)], dtype=dtype)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        outs = gen.infer(xs[i % len(xs)], dtype=dtype)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(outs).all())
    if rank == 0:
        ms = 1e3 * dt / args.steps
        gflop = INFER_GFLOP_PER_SAMPLE_64 * (T / 64.0) * B
        peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS
        # roofline of the DOMINANT kernel family of one traced forward, from the launchers' own EXECUTED FLOPs (the fp32 path runs its
        # 5x5 layers as Winograd products: 0.36x / 0.44x of the direct-convolution FLOPs, so "algorithmic FLOPs / time" can exceed the
        # MFMA peak and is reported separately, not as a roofline fraction)
        from mask_cyclegan_vc import _hip
        L = _hip.lib()
        L.mcvc_trace_kind_name.restype = ctypes.c_char_p
        nk = L.mcvc_trace_kinds()
        buf = (ctypes.c_double * (4 * nk))()
        torch.cuda.synchronize()
        L.mcvc_trace_enable(1)
        gen.infer(xs[0], dtype=dtype)
        L.mcvc_trace_collect(buf)
        L.mcvc_trace_enable(0)
        rows = [{"kernel": L.mcvc_trace_kind_name(k).decode(), "launches": int(buf[4 * k]), "ms": buf[4 * k + 1], "gflop": buf[4 * k + 2] / 1e9}
                for k in range(nk) if buf[4 * k] > 0]
        rows.sort(key=lambda r: -r["ms"])
        kt = sum(r["ms"] for r in rows)
        dom = rows[0]
        ach = dom["gflop"] / dom["ms"]
        res = {"metric": "generator_A2B inference mel-frames/s, bs=%d x %d frames" % (B, T), "value": world * B * T * args.steps / dt,
               "unit": "mel-frames/s (= batch x frames / latency, summed over GPUs)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": "generator_A2B forward (test.py path), bs=%d, 80 mel x %d frames, all-ones mask, %s, default-init "
                                      "weights (seed 0)" % (B, T, dtype), "global_batch": world * B, "parallelism": "replicas%d" % world},
               "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                            "kernel": dom["kernel"], "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                            "share_of_kernel_time": dom["ms"] / kt, "flops": "executed (as launched)"},
               "kernel_time_ms_per_step": {r["kernel"]: round(r["ms"], 4) for r in rows},
               "executed_gflop_per_step": round(sum(r["gflop"] for r in rows), 1),
               "algorithmic_gflop_per_step": round(gflop, 1),               # direct-convolution FLOPs of the forward
               "algorithmic_tflops": gflop / ms,
               "outputs_finite": finite}
        if world == 1 and args.cpu_iters != 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import mcvc_oracle as orc
            torch.set_num_threads(args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 32))
            params = {k: v.detach().cpu() for k, v in gen.state_dict().items()}
            nb = 2                                       # bounded sample: two samples of the batch (a full batch is ~16x that)
            x = xs[0][:nb].cpu()
            with torch.no_grad():
                orc.generator_forward(params, x[:1], torch.ones_like(x[:1]))      # warm-up
                t0 = time.perf_counter()
                ref = orc.generator_forward(params, x, torch.ones_like(x))
                cdt = time.perf_counter() - t0
            got = gen.infer(xs[0], dtype=dtype)[:nb].float().cpu()
            res["cpu_baseline"] = {"value": nb * T / cdt, "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                                   "sample": "oracle.generator_forward (fp32) on %d of the %d samples x %d frames (%.1f s)" % (nb, B, T, cdt)}
            res["parity_vs_cpu_rel_l2"] = float((got.double() - ref.double()).norm() / ref.double().norm())
            res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        return res
    return None


_T0 = time.perf_counter()


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before the timed region (default 10; --mode infer: 100 -- the HIP runtime "
                    "stalls ONCE for 10-50 ms somewhere between ~400 and ~3000 launches into a process, and with 34 launches per forward a 10-forward "
                    "warm-up put that stall inside or outside the 50 timed forwards depending on the launch count: measured r6, profiles/r06_c2d1d_warmup_sweep.log)")
    ap.add_argument("--mode", choices=("train", "infer"), default="train", help="train: full G+D iteration (BASELINE metric); "
                    "infer: generator_A2B forward, bs=16 x 512 frames (BASELINE configs[4])")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None, help="infer mode arithmetic (default bf16); training is fp32")
    ap.add_argument("--n-batches", type=int, default=64, help="pre-generated synthetic minibatches cycled through (SURVEY.md section 8d)")
    ap.add_argument("--deterministic", action="store_true", help="bit-reproducible mode (no floating-point atomics)")
    ap.add_argument("--batch-size", type=int, default=1, help="per-GPU minibatch (BASELINE metric: 1)")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--cpu-iters", type=int, default=-1, help="timed CPU-baseline iterations (0 = skip; default: about 10-30 s of CPU work)")
    ap.add_argument("--no-trace", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="default single-GPU run: skip the nested records of BASELINE configs[2..4]")
    ap.add_argument("--sync-losses", action="store_true", help="read the CURRENT iteration's g_loss and d_loss on the host every step, as the "
                    "reference's .item() calls do (train.py:302-304): completes the iteration before the next one is issued (no pipelining)")
    ap.add_argument("--loss-lag", type=int, default=2, help="the per-step loss readback returns the iteration issued this many step()s earlier "
                    "(2 = the training loop's LOSS_LAG: never waits for work in flight; 1 = rounds 3-4: waits for the discriminator phase just queued)")
    ap.add_argument("--allow-degraded", action="store_true", help="--gpus N: do not fail when the ranks did not all take part in the collective or the "
                    "persistent trunk kernels fell back to per-layer launches (single-GPU choreography tests over gloo)")
    ap.add_argument("--rccl-one-rank", action="store_true", help="single-GPU self-check of the data-parallel path: a ONE-rank nccl (= RCCL) group with the "
                    "gradient exchange forced on -- real RCCL kernels on the communication stream beside the rank schedule; reports "
                    "exposed_comm_ms_per_step like a --gpus N run (n_gpus stays 1)")
    ap.add_argument("--test-force-residency", type=int, default=None, help=argparse.SUPPRESS)    # tests: engines claim this many persistent passes in flight
    ap.add_argument("--test-distinct-gpus", action="store_true", help=argparse.SUPPRESS)         # tests: skip the ranks-share-a-GPU switch
    ap.add_argument("--serial", action="store_true", help="one stream: no lanes, no auxiliary weight-gradient stream (A/B comparison, per-kernel profiling)")
    ap.add_argument("--dump-trace", default=None, help="write one traced step's per-launch records (launch order) to this file")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = min(host cores, 32))")
    a = ap.parse_args()
    if a.warmup is None:
        a.warmup = 100 if a.mode == "infer" else 10
    return a


def dist_info(world, device, force=False):
    """What the gradient exchange actually ran on: an all-reduce of ones proves that `world` ranks took part in a collective on the
    benchmark's own backend (RCCL when one GPU per rank is visible)."""
    if world <= 1 and not force:
        return {"backend": None, "world": 1, "rccl_ranks_seen": 1}
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    return {"backend": dist.get_backend(), "world": world, "rccl_ranks_seen": int(round(float(ones.item()))),
            "gpus_visible": torch.cuda.device_count()}


def train_record(args, rank, world, device, B, T, steps, warmup, cpu_iters, n_batches, cpu_warm=True, config_id=None, stop_identity_after=1e4):
    """One timed training configuration (a full G+D iteration at per-GPU batch B): W warm-up + exactly K timed steps between
    barrier + synchronize pairs, max over ranks; then one traced step (per-kernel HIP events) for the roofline of the dominant kernel
    family and the CPU oracle on a bounded sample of the same workload.  Returns the record on rank 0 (None elsewhere)."""
    from mask_cyclegan_vc.engine import TrainEngine
    from mask_cyclegan_vc.parallel import FlatGradReducer
    from mask_cyclegan_vc.schedule import StepSchedule
    log("bs=%d: building nets" % B)
    nets = build_nets(device)
    sched = StepSchedule(generator_lr=2e-4, discriminator_lr=1e-4, num_epochs=6172, n_samples=81, batch_size=B,
                         decay_after=2e5, stop_identity_after=stop_identity_after, world_size=world)     # bash_scripts/mask_cyclegan_train.sh
    one_rank = bool(getattr(args, "rccl_one_rank", False))
    dist_on = world > 1 or one_rank
    reducer = FlatGradReducer(force=one_rank)
    engine = TrainEngine(nets, B, T, schedule=sched, reducer=reducer)
    engine.concurrent = not args.serial
    if args.serial:
        engine.aux_wgrad = False          # truly one stream: per-kernel durations comparable with the traced step's
    batches = synthetic_batches(n_batches, B, T, rank, device)
    log("engine ready; warm-up")

    first = None
    for i in range(warmup):
        engine.step(*batches[i % len(batches)])
        if i == 0:
            lo = engine.losses()
            first = (lo["g_loss"], lo["d_loss"])
    engine.flush()                         # a deferred (data-parallel) discriminator update belongs to the warm-up
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    log("timing %d steps" % steps)
    reducer.time_waits = dist_on         # exposed (un-hidden) gradient-exchange time: an event pair around every wait for the communication stream
    t0 = time.perf_counter()
    sync_losses = bool(getattr(args, "sync_losses", False))
    loss_lag = max(1, int(getattr(args, "loss_lag", 2)))
    host_enq = host_wait = 0.0             # host seconds inside step() (enqueueing ~300 launches) and inside the loss readback (waiting for the GPU)
    for i in range(steps):
        h0 = time.perf_counter()
        engine.step(*batches[(warmup + i) % len(batches)])
        h1 = time.perf_counter()
        # the reference reads both losses every iteration (train.py:303).  So do we -- the losses of a COMPLETE iteration, two step()s
        # behind, exactly as mask_cyclegan_vc/train.py's loop does (LOSS_LAG): the newest complete iteration's discriminator phase was
        # queued by THIS step() (it runs beside this iteration's generator phase), so waiting for it drains the GPU before the host
        # issues the next iteration (r4 / early r5: +0.3 ms per bs=1 step of idle GPU); the one before it was published a step() ago.
        # --sync-losses: the reference-exact readback -- THIS iteration's pair, which completes the iteration first
        engine.losses(lagged=0 if sync_losses else loss_lag)
        host_enq += h1 - h0; host_wait += time.perf_counter() - h1
    engine.flush()                         # ... and the last one to the timed region: exactly K complete iterations
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    log("timed region done: %.2f ms/step" % (1e3 * dt / steps))
    reducer.time_waits = False
    exposed_ms, n_waits = reducer.exposed_ms()
    if dist_on:
        t = torch.tensor([exposed_ms], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = float(t.item())
    engine.check_faults()
    final = engine.losses()
    finite = all(np.isfinite(v) for v in final.values())
    rows, raw = ([], []) if args.no_trace else trace_one_step(engine, batches)
    if args.dump_trace and rank == 0 and config_id is None and raw:
        with open(args.dump_trace, "w") as fh:
            for i, (k, ms, fl, by) in enumerate(raw):
                fh.write("%4d %-24s %9.4f ms %10.4f GF %9.3f MB %8.2f TF/s %8.1f GB/s\n" % (
                    i, k, ms, fl / 1e9, by / 1e6, fl / 1e9 / max(ms, 1e-6), by / 1e6 / max(ms, 1e-6)))
    host = {"enqueue_ms_per_step": 1e3 * host_enq / steps, "loss_wait_ms_per_step": 1e3 * host_wait / steps,
            "note": "host wall time inside engine.step() / inside the loss readback; the step is GPU-bound while enqueue << ms_per_step"}
    schedule = {"grouped_launches": bool(engine._use_grouped()), "pipelined": bool(engine._use_pipeline()), "merged_forwards": bool(engine._use_merged()),
                "queue_probe": getattr(engine, "queue_probe", None),
                "loss_readback": ("both losses of the CURRENT iteration every step (reference-exact, train.py:302-304): the pipelined overlap of "
                                  "iteration t's discriminator phase with iteration t+1's generator phase is given up") if sync_losses else
                                 ("both losses every iteration, those of the iteration issued two step()s earlier (complete; the read never waits "
                                  "for work in flight) -- the same readback as the training loop's (train.py LOSS_LAG)" if loss_lag >= 2 else
                                  "both losses every iteration, those of the last complete iteration (one step() behind: waits for the discriminator phase just queued)"),
                "trunk_persistent": engine.L.mcvc_gen_trunk_persistent(B, T) if engine._use_grouped() else None,
                "trunk_fallback": bool(engine.trunk_fallback)}
    # what the persistent trunk kernels would do with ONE pass in flight: tells "this shape has no persistent kernel" from "the residency
    # bound of this schedule switched them off" (the silent degradation a multi-GPU line must not hide)
    engine.L.mcvc_set_trunk_passes_in_flight(1)
    schedule["trunk_persistent_possible"] = engine.L.mcvc_gen_trunk_persistent(B, T) if engine._use_grouped() else None
    engine._resid = None
    engine._set_residency()
    del engine, nets
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    ms = 1e3 * dt / steps
    value = world * steps / dt
    sample_iters = world * B * steps / dt
    res = {
        "metric": "train iters/s (full G+D step), 80x%d mel bs=%d" % (T, B),
        "value": value, "unit": "iters/s (per-GPU bs=%d iterations, summed over GPUs)" % B,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "MaskCycleGAN-VC full G+D iteration, VCC2018-shaped synthetic mels, bs=%d/GPU, 80 mel x %d frames, fp32, "
                               "default-init weights (seed 0)" % (B, T),
                   "global_batch": world * B, "parallelism": "dp%d" % world},
        "mel_frames_per_s": sample_iters * T,
        "step_mfma_fraction": sample_iters * ALG_GFLOP_PER_SAMPLE_ITER / 1e3 / (PEAK_FP32_MFMA_TFLOPS * world),
        "exposed_comm_ms_per_step": (exposed_ms / steps) if dist_on else 0.0, "comm_waits_per_step": n_waits / steps,
        "losses_finite": finite, "last_losses": final, "schedule": schedule, "host": host, "n_batches": len(batches), "deterministic": bool(args.deterministic),
        "identity_loss_lambda": float(sched.identity_loss_lambda),
    }
    if config_id:
        res["config_id"] = config_id
    if rows:
        conv = [r for r in rows if r["gflop"] > 0]
        dom = max(conv, key=lambda r: r["ms"])
        total_ms = sum(r["ms"] for r in rows)
        ach = dom["gflop"] / dom["ms"]          # GFLOP/ms == TFLOP/s
        traffic, traffic_src = pmc_traffic(dom["kernel"], B)
        res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "kernel": dom["kernel"],
                           "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                           "share_of_kernel_time": dom["ms"] / total_ms, "flops": "executed (as launched)",
                           "executed_gflop_per_step": round(dom["gflop"], 2)}
        res["kernel_time_ms_per_step"] = {r["kernel"]: round(r["ms"], 4) for r in rows}
        res["kernel_launches_per_step"] = int(sum(r["launches"] for r in rows))
        res["all_conv_tflops"] = sum(r["gflop"] for r in conv) / sum(r["ms"] for r in conv)
        # step_mfma_fraction counts the ALGORITHMIC (direct-convolution) FLOPs of SURVEY 8d; the 5x5 layers execute 0.36x / 0.44x of
        # theirs as Winograd products, so at large batch that figure approaches (and may pass) 1 -- the executed one cannot
        res["executed_gflop_per_sample_iter"] = round(sum(r["gflop"] for r in conv) / B, 1)
        res["algorithmic_gflop_per_sample_iter"] = ALG_GFLOP_PER_SAMPLE_ITER
        res["step_executed_mfma_fraction"] = sum(r["gflop"] for r in conv) / ms / PEAK_FP32_MFMA_TFLOPS
        # SURVEY 8(d) "conv-roofline time" of the step AS EXECUTED: every launch is bound by the matrix pipe or by HBM, whichever is slower
        # for the FLOPs it executes (Winograd products count their own multiplies, not the direct convolution's) and the bytes its launcher
        # counts (operands + results incl. the Winograd-domain U / V / M tensors, slabs, packed copies): sum over launches of
        # max(FLOPs / 157.3 TF/s, bytes / 8 TB/s).  frac_of_conv_roofline = that floor / the measured step time.
        roof_ms = sum(max(fl / (PEAK_FP32_MFMA_TFLOPS * 1e12), by / (PEAK_HBM_GBS * 1e9)) for _k, _ms, fl, by in raw) * 1e3
        res["conv_roofline_ms"] = round(roof_ms, 4)
        res["frac_of_conv_roofline"] = roof_ms / ms
        # two byte counts, named for what they are: what the LAUNCHERS count for the kernels of the traced step (operands + results as the
        # host code knows them: live), and what the HBM counters saw (PMC passes of the named revision, per kernel instance: a constant)
        res["hbm_bytes_per_step_launcher"] = sum(by for _k, _ms, _fl, by in raw)
        pmc_b, pmc_fam, pmc_src = pmc_bytes_per_step(B)
        res["hbm_bytes_per_step_pmc"] = pmc_b
        res["hbm_bytes_per_step_pmc_source"] = pmc_src
        alg_b = ALG_BYTES_PER_ITER.get(B)
        res["algorithmic_bytes_per_step"] = alg_b                                # SURVEY 8(d) lower bound incl. Adam (None: not tabulated)
        if alg_b:
            res["hbm_bytes_ratio_to_algorithmic"] = {"launcher": res["hbm_bytes_per_step_launcher"] / alg_b,
                                                     "pmc": (pmc_b / alg_b) if pmc_b else None}
        if pmc_fam:
            launcher_fam = {r["kernel"]: r["mbytes"] * 1e6 for r in rows}
            res["hbm_bytes_per_step_by_family"] = {k: {"pmc": round(v), "launcher": round(launcher_fam.get(k, 0.0))} for k, v in list(pmc_fam.items())[:12]}
        res["serial_kernel_ms_per_step"] = round(total_ms, 4)
    log("trace done")
    if world == 1 and cpu_iters > 0:
        cb, parity = cpu_baseline(B, T, cpu_iters, first, args.cpu_threads, warm=cpu_warm)
        res["cpu_baseline"] = cb
        res["parity_first_iteration_vs_cpu"] = parity
        res["speedup_vs_cpu"] = value / cb["value"]
    return res


def main():
    args = parse_args()
    from mask_cyclegan_vc import parallel
    from mask_cyclegan_vc.parallel import init_from_env
    if args.test_distinct_gpus:
        parallel.ASSUME_DISTINCT_GPUS = True
    if args.test_force_residency is not None:
        from mask_cyclegan_vc.engine import TrainEngine
        TrainEngine.FORCE_INFLIGHT = args.test_force_residency

    rank, world, local_rank = init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    local_rank %= max(torch.cuda.device_count(), 1)      # (only differs on a box with fewer GPUs than ranks: gloo test runs)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.deterministic:
        from mask_cyclegan_vc import _hip
        _hip.lib().mcvc_set_deterministic(1)
    if args.rccl_one_rank:
        if world != 1 or args.mode != "train":
            raise SystemExit("--rccl-one-rank is a single-process training self-check")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        args.no_extra_configs = True
    if args.mode == "infer":
        res = infer_record(args, rank, world, device, args.dtype or "bf16", args.batch_size if args.batch_size != 1 else 16,
                           args.frames if args.frames != 64 else 512, args.steps, args.warmup)
    else:
        B, T = args.batch_size, args.frames
        cpu_iters = args.cpu_iters
        if cpu_iters < 0:                  # bounded CPU sample: ~1 s per bs=1 iteration on 32 threads of the GPU box's host
            cpu_iters = max(2, min(12, 12 // B))
        res = train_record(args, rank, world, device, B, T, args.steps, args.warmup, cpu_iters, args.n_batches)
        # The default single-GPU invocation (the one the driver runs) also measures the other BASELINE configs, each as a nested record
        # with its own roofline and CPU sample: configs[2] = the bs=32 step, the per-GPU shape of configs[3] = the bs=8 step, configs[4] =
        # generator_A2B bf16 inference at 16 x 512 frames.  The headline line above them is unchanged (configs[1], bs=1).
        if world == 1 and B == 1 and T == 64 and not args.no_extra_configs and not args.serial:
            extra = []
            sub = argparse.Namespace(**dict(vars(args), dump_trace=None))
            skip_cpu = args.cpu_iters == 0
            extra.append(train_record(sub, rank, world, device, 32, 64, 8, 3, 0 if skip_cpu else 1, 8, cpu_warm=False,
                                      config_id="configs[2]: bs=32, 1 GPU"))
            extra.append(train_record(sub, rank, world, device, 8, 64, 20, 5, 0 if skip_cpu else 2, 16, cpu_warm=False,
                                      config_id="configs[3] per-GPU shape: bs=8 (the 8-GPU run itself is the driver's)"))
            sub_i = argparse.Namespace(**dict(vars(sub), cpu_iters=0 if skip_cpu else -1))
            rec = infer_record(sub_i, rank, world, device, "bf16", 16, 512, 30, 5)
            if rec is not None:
                rec["config_id"] = "configs[4]: generator_A2B inference, bs=16 x 512 frames, bf16"
            extra.append(rec)
            if res is not None:
                res["configs"] = [r for r in extra if r is not None]
        # the price of the reference-exact loss readback, driver-observed: the same config once more, short, with
        # the CURRENT iteration's losses read every step
        if world == 1 and B == 1 and T == 64 and not args.no_extra_configs and not (args.serial or args.sync_losses):
            sub = argparse.Namespace(**dict(vars(args), dump_trace=None, no_trace=True, sync_losses=True))
            rec = train_record(sub, rank, world, device, 1, 64, 30, 5, 0, 16, config_id="sync-losses")
            if res is not None and rec is not None:
                res["schedule"]["sync_losses_ms_per_step"] = rec["ms_per_step"]
                res["schedule"]["sync_losses_cost"] = rec["ms_per_step"] / res["ms_per_step"] - 1.0
            # ... and the regime a canonical run (bash_scripts/mask_cyclegan_train.sh: identity loss until 1e4 samples of ~5e5) spends > 97 % of
            # its iterations in: identity_loss_lambda = 0 (train.py:314-315), where the identity passes weigh nothing and are not computed.
            # The HEADLINE above stays the more expensive regime before the cut-off (lambda = 5): this is a second, labelled number.
            sub = argparse.Namespace(**dict(vars(args), dump_trace=None, no_trace=True))
            rec = train_record(sub, rank, world, device, 1, 64, 40, 8, 0, 16, config_id="post-cutoff", stop_identity_after=0)
            if res is not None and rec is not None:
                res["after_identity_cutoff"] = {"ms_per_step": rec["ms_per_step"], "iters_per_s": rec["value"], "identity_loss_lambda": rec["identity_loss_lambda"],
                                                "note": "same config with identity_loss_lambda = 0 (train.py:314-315): > 97 % of a canonical run's iterations"}
    info = dist_info(world, device, force=args.rccl_one_rank)
    degraded = []
    if world > 1 and args.mode == "train":
        if info["rccl_ranks_seen"] != world:
            degraded.append("collective saw %d of %d ranks" % (info["rccl_ranks_seen"], world))
        sc = (res or {}).get("schedule", {}) if rank == 0 else {}
        if sc.get("trunk_fallback") or (sc.get("trunk_persistent_possible") and not sc.get("trunk_persistent")):
            degraded.append("persistent trunk kernels fell back to per-layer launches on the ranks (residency bound or fault)")
    if rank == 0 and res is not None:
        res["dist"] = info
        if degraded:
            res["degraded"] = degraded
        print(json.dumps(res))
    if args.rccl_one_rank:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    if world > 1:
        flag = torch.tensor([1.0 if degraded else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
        if float(flag.item()) > 0 and not args.allow_degraded:
            # a SCALE line must not be quietly degraded: the line above says what happened, the exit code makes the run fail
            log("DEGRADED multi-GPU run: %s" % "; ".join(degraded))
            sys.exit(3)


if __name__ == "__main__":
    main()
