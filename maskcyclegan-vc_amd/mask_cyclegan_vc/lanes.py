"""Lane streams of the training engine: HIP streams on distinct hardware queues, one set per (device, caller stream) and process."""
from __future__ import annotations

import torch

_LANES = {}      # (device, caller stream) -> (legacy sides, aux streams, probed sides, probe report)


def pick_side_streams(dev, want=3):
    """Three side streams that share a hardware queue neither with each other nor with the caller's stream.

    ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and two streams on one queue serialise: a
    kernel on one waits for everything queued earlier on the other -- a false dependency between lanes that the task graph does
    not contain (tools/queue_probe.py prints the classes; PyTorch's pooled streams land on the queues in no simple order).  So the
    lanes are CHOSEN: candidates are probed against the streams already picked (a ~0.25 ms spin on one, a tiny kernel on the other;
    the tiny kernel finishing only with the spin = same queue) and kept when independent."""
    cands = [torch.cuda.Stream(device=dev) for _ in range(16)]
    main = torch.cuda.current_stream(dev)
    x = torch.zeros(256, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    spin = 2_000_000
    torch.cuda.synchronize(dev)
    e0.record(); torch.cuda._sleep(spin); e1.record(); torch.cuda.synchronize(dev)
    spin = max(1000, int(spin * 0.25 / max(e0.elapsed_time(e1), 1e-3)))
    e0.record(); torch.cuda._sleep(spin); e1.record(); torch.cuda.synchronize(dev)
    spin_ms = e0.elapsed_time(e1)

    def delays(a, b):          # does a spin on stream a hold back a kernel queued afterwards on stream b?
        torch.cuda.synchronize(dev)
        start, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(a):
            start.record(a)
            torch.cuda._sleep(spin)
        with torch.cuda.stream(b):
            x.add_(1.0)
            done.record(b)
        torch.cuda.synchronize(dev)
        return start.elapsed_time(done) > 0.6 * spin_ms
    picked = []
    for c in cands:
        # (work queued on ANY stream after null-stream work waits for it -- the legacy default-stream rule -- so against a null main
        # stream only the other direction identifies a shared queue)
        indep = delays(c, main) is False and (main.cuda_stream == 0 or not delays(main, c))
        indep = indep and all(not delays(c, p) and not delays(p, c) for p in picked)
        if indep:
            picked.append(c)
            if len(picked) == want:
                break
    report = {"independent_lanes": len(picked), "spin_ms": spin_ms, "main_is_null_stream": main.cuda_stream == 0}
    for c in cands:                # fewer than `want` independent queues (GPU_MAX_HW_QUEUES < 4): fill up with what there is
        if len(picked) < want and c not in picked:
            picked.append(c)
    return picked, report


def get_lanes(dev):
    """(legacy side lanes, auxiliary streams, probed side lanes, probe report) for the caller's current stream on ``dev``.

    Two sets of side lanes: the four-lane schedule keeps the streams (and the auxiliary streams created right behind them) it was tuned
    on -- with probed lanes its auxiliary streams land on the lanes' queues: bs=8 30.5 -> 32.3 ms -- the grouped / pipelined schedule uses
    lanes probed onto distinct hardware queues.  One set per (device, caller stream) and process: a second engine in the same process
    (bench.py's extra batch sizes, an evaluation engine beside the training one) reuses them -- every new HIP stream lands on one of the 4
    hardware queues in pool order, and a later engine's fresh streams landed on worse combinations (nested bs=8 24.1 ms against 22.6 in a
    process of its own)."""
    key = (torch.device(dev).index or 0, torch.cuda.current_stream(dev).cuda_stream)
    lanes = _LANES.get(key)
    if lanes is None:
        legacy = [torch.cuda.Stream(device=dev) for _ in range(3)]
        # inside a backward pass the weight-gradient kernels are off the critical path: one auxiliary stream per lane
        aux = [torch.cuda.Stream(device=dev) for _ in range(4)]
        # (a stream gets its hardware queue at first USE: touch these in creation order before the probe puts work on its candidates)
        _t = torch.zeros(64, device=dev)
        for st in legacy + aux:
            with torch.cuda.stream(st):
                _t.add_(1.0)
        torch.cuda.synchronize(dev)
        probed, report = pick_side_streams(dev)
        lanes = _LANES[key] = (legacy, aux, probed, report)
    return lanes
