"""world_size-2 gloo test of the data-parallel gradient exchange (mask_cyclegan_vc/parallel.py) on CPU.

Claim under test: summing per-rank flat gradients and scaling by 1/R reproduces the
gradient of the single-process step on the concatenated batch, because every op of the model is per-sample and
every loss is a batch mean.  The per-rank gradients come from the CPU oracle (test infrastructure); the reducer is
product code and is device-agnostic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_d_grads(orc, params, x):
    names = [k for k in orc.discriminator_param_names() if not k.startswith(orc.DISC_DEAD_PREFIX)]
    leaves = [params[k].requires_grad_(True) for k in names]
    loss = torch.mean((1 - orc.discriminator_forward(params, x)) ** 2)
    grads = torch.autograd.grad(loss, leaves)
    return torch.cat([g.reshape(-1) for g in grads]), float(loss)


def _worker(rank, world, port, out_path):
    for sub in ("oracle", "maskcyclegan-vc_amd"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    import mcvc_oracle as orc
    from mask_cyclegan_vc.parallel import FlatGradReducer, init_from_env
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    params = orc.filler_params("D", 7)
    x = torch.from_numpy(np.random.RandomState(100 + rank).randn(1, 80, 64).astype(np.float32))
    flat, _ = _flat_d_grads(orc, params, x)
    red = FlatGradReducer(bucket_bytes=1 << 20)           # 1 MiB buckets -> ~24 collectives, exercises the bucketing
    assert red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
    red.reduce_(flat)
    flat *= red.grad_scale
    # parameters broadcast from rank 0
    p = torch.full((1000,), float(rank))
    red.broadcast_(p, src=0)
    assert float(p.abs().max()) == 0.0
    if rank == 0:
        torch.save(flat, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_flat_allreduce_equals_large_batch_gradient(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mcvc_oracle as orc
    out = str(tmp_path / "reduced.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    reduced = torch.load(out)
    params = orc.filler_params("D", 7)
    xs = [torch.from_numpy(np.random.RandomState(100 + r).randn(1, 80, 64).astype(np.float32)) for r in range(2)]
    ref, _ = _flat_d_grads(orc, params, torch.cat(xs, 0))
    rel = float((reduced.double() - ref.double()).norm() / ref.double().norm())
    assert rel < 1e-5, rel
    assert reduced.numel() == 6202881          # live discriminator parameters


def test_single_process_reducer_is_identity():
    for sub in ("maskcyclegan-vc_amd",):
        sys.path.insert(0, os.path.join(ROOT, sub))
    from mask_cyclegan_vc.parallel import FlatGradReducer
    red = FlatGradReducer()
    t = torch.arange(10.0)
    assert red.world == 1 and red.grad_scale == 1.0
    assert torch.equal(red.reduce_(t.clone()), t)
