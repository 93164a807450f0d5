"""Data-parallel gradient exchange for the MaskCycleGAN-VC step (new work: the reference has no
distributed code at all -- SURVEY.md section 2 "Parallelism" / section 8e).

One process per GPU (torchrun).  Samples are independent end to end (InstanceNorm is per sample,
every loss is a batch mean), so averaging per-rank gradients over R equal ranks IS the gradient of
the single-process step with batch R*b.  Gradients already live in two flat fp32 buffers (all
generator tensors: 196.3 MB; live discriminator tensors: 99.2 MB), so the exchange is one
bucketed in-place all-reduce per optimizer phase with no packing copies; the 1/R scale is folded
into the fused Adam kernel (``grad_scale``).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): ring all-reduce is per-link bound, so buckets
are large (default 64 MiB) -- enough of them to pipeline behind each other on the communication
stream, few enough to amortise the per-collective launch latency.

The reducer is device-agnostic torch.distributed code (RCCL = backend "nccl" on ROCm, gloo on CPU
for the world_size>1 unit tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# Multi-process GPU work on this image needs dmabuf IPC: export HSA_ENABLE_IPC_MODE_LEGACY=0 in the LAUNCHER's environment (RCCL fails in
# hipIpcGetMemHandle without it; the HSA runtime reads it when it starts, so setting it here would be too late -- INTEGRATION.md).


ASSUME_DISTINCT_GPUS = False      # test hook (bench.py --test-distinct-gpus): skip the shared-device switch below


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MCVC_DIST_BACKEND=gloo: test hook (two ranks sharing the one GPU of a dev box; RCCL refuses duplicate devices)
            backend = os.environ.get("MCVC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            # RCCL needs one GPU per rank: fail loudly instead of hanging in the first collective
            if torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
                raise RuntimeError("RCCL (backend nccl) needs one GPU per local rank: %d visible, %s ranks on this node "
                                   "(set MCVC_DIST_BACKEND=gloo only for single-GPU choreography tests)"
                                   % (torch.cuda.device_count(), os.environ.get("LOCAL_WORLD_SIZE", world)))
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            if torch.cuda.is_available() and _ranks_share_a_gpu(local_rank) and not ASSUME_DISTINCT_GPUS:
                # Ranks SHARING a GPU (the single-GPU choreography tests only): the persistent trunk kernels wait inside the kernel for
                # workgroups that must all be resident (csrc/trunk.h) -- that holds for the passes one process keeps in flight, not for
                # two processes' worth of them on one device.  Run the trunk as per-layer launches there.
                from . import _hip
                _hip.lib().mcvc_set_trunk_persistent(0)
                if rank == 0:
                    import sys
                    print("[mcvc] ranks share a GPU: persistent trunk kernels off (per-layer launches)", file=sys.stderr, flush=True)
    return rank, world, local_rank


def _device_identity(local_rank):
    """(host, physical device) of this rank.  The LOGICAL index is not an identity: a launcher that hands every rank one GPU through
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES makes it 0 on all of them.  The device's UUID (or its PCI address) is."""
    import socket
    ndev = max(torch.cuda.device_count(), 1)
    idx = local_rank % ndev
    ident = None
    try:
        props = torch.cuda.get_device_properties(idx)
        uuid = getattr(props, "uuid", None)
        if uuid is not None and str(uuid).strip("0-") != "":
            ident = "uuid:%s" % uuid
        elif hasattr(props, "pci_bus_id"):
            ident = "pci:%s:%s:%s" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    except Exception:        # noqa: BLE001 -- an identity probe must never take the job down; fall back to the logical index
        ident = None
    if ident is None:
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or ""
        ident = "idx:%s:%d" % (vis, idx)
    return (socket.gethostname(), ident)


def _ranks_share_a_gpu(local_rank):
    """True when two ranks of the job run on the same physical device -- from the ACTUAL rank -> (host, device identity) map (an
    all-gather), not from LOCAL_WORLD_SIZE, which a launcher other than torchrun may not set (a 16-rank / 2-node job would otherwise be
    mis-read as 16 ranks on 8 GPUs), and not from the logical device index (see _device_identity)."""
    pairs = [None] * dist.get_world_size()
    dist.all_gather_object(pairs, _device_identity(local_rank))
    return len(set(pairs)) < len(pairs)


class FlatGradReducer:
    """Sum-all-reduce a flat gradient buffer in place, in buckets, optionally on a side stream."""

    def __init__(self, bucket_bytes=64 << 20, group=None, use_side_stream=True, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # ``active``: the exchange is really issued.  ``force`` keeps it on in a ONE-rank group (tests/test_hip_rccl_one_rank.py: real RCCL
        # kernels on the communication stream beside the persistent trunk kernels, the data-parallel schedule and its residency budget, on
        # the single GPU of a dev box -- a 1-rank all-reduce is the identity, so the result must equal the run without a reducer bit for bit)
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._stream = None
        self._use_side_stream = use_side_stream
        # bench.py --gpus N: time every wait of a compute stream for the communication stream (an event pair around the wait on the
        # WAITING stream: what it measures is the time that stream sat blocked, i.e. the communication that was not hidden)
        self.time_waits = False
        self._wait_events = []

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def reduce_(self, flat: torch.Tensor):
        """In-place SUM over ranks (scale by ``grad_scale`` downstream). No-op for world size 1."""
        if not self.active:
            return flat
        n = flat.numel()
        if flat.is_cuda and self._use_side_stream:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            cur = torch.cuda.current_stream(flat.device)
            self._stream.wait_stream(cur)           # gradients are complete on the compute stream
            with torch.cuda.stream(self._stream):
                for lo in range(0, n, self.bucket_elems):
                    dist.all_reduce(flat[lo:min(n, lo + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group)
            self._wait_on(cur)                      # Adam (compute stream) consumes the reduced buffer
        else:
            for lo in range(0, n, self.bucket_elems):
                dist.all_reduce(flat[lo:min(n, lo + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def reduce_async_(self, flat: torch.Tensor):
        """Start the bucketed SUM all-reduce of ``flat`` on the communication stream and return immediately; the caller's
        stream keeps computing.  ``wait()`` makes the current stream wait for it.  (The engine uses this to hide the
        discriminator gradient exchange behind the next iteration's generator forwards.)"""
        if not self.active:
            return flat
        if not (flat.is_cuda and self._use_side_stream):
            return self.reduce_(flat)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        cur = torch.cuda.current_stream(flat.device)
        self._stream.wait_stream(cur)
        n = flat.numel()
        with torch.cuda.stream(self._stream):
            for lo in range(0, n, self.bucket_elems):
                dist.all_reduce(flat[lo:min(n, lo + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def reduce_range_after_(self, flat: torch.Tensor, lo: int, hi: int, event=None):
        """Queue the SUM all-reduce of ``flat[lo:hi]`` on the communication stream, to start once ``event`` (recorded by the
        producer of that range; None = everything queued on the current stream so far) has completed.  Pair with ``wait()``."""
        if not self.active or hi <= lo:
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        if event is not None:
            self._stream.wait_event(event)
        else:
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(self._stream):
            for a in range(lo, hi, self.bucket_elems):
                dist.all_reduce(flat[a:min(hi, a + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group)

    def _wait_on(self, cur):
        if self.time_waits:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            cur.wait_stream(self._stream)
            e1.record(cur)
            self._wait_events.append((e0, e1))
        else:
            cur.wait_stream(self._stream)

    def wait(self, device=None):
        """Order the current stream after everything queued on the communication stream."""
        if self._stream is not None:
            self._wait_on(torch.cuda.current_stream(device))

    def exposed_ms(self):
        """(total milliseconds compute streams spent blocked in ``wait``, number of waits) since the last call; synchronises the device."""
        if not self._wait_events:
            return 0.0, 0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        n = len(self._wait_events)
        self._wait_events = []
        return ms, n

    def broadcast_(self, flat: torch.Tensor, src=0):
        """Make every rank start from rank ``src``'s parameters."""
        if self.active:
            dist.broadcast(flat, src=src, group=self.group)
        return flat
