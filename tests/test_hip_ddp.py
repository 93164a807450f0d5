"""GPU: the data-parallel engine with two real processes, on both transports.

* ``gloo`` -- a dev box has ONE GPU, so both ranks share cuda:0 and exchange over gloo (RCCL refuses duplicate devices); what is under
  test is the engine's choreography: flat-buffer all-reduce, 1/R folded into Adam, range reductions behind milestone events, the deferred
  discriminator update on the communication stream.
* ``nccl`` (= RCCL over xGMI) -- one GPU per rank, the production transport; runs wherever two GPUs are visible and is skipped (not
  failed) elsewhere.  Same worker, same assertions: averaged per-rank gradients == one process on the concatenated batch, bit-identical
  parameters on all ranks after three full iterations, 2-rank mean losses == single-process losses."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(backend, port, per_rank=8, **extra):
    env = dict(os.environ, MCVC_TEST_DDP_BACKEND=backend, MCVC_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MCVC_TEST_DDP_PER_RANK=str(per_rank), **extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "ddp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "parameter spread across ranks 0.000e+00" in r.stdout
    assert ("backend %s" % backend) in r.stdout
    return r.stdout


def test_two_ranks_match_one_process_and_each_other():
    _run("gloo", 29533)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has fewer than two)")
def test_two_ranks_over_rccl_one_gpu_each():
    _run("nccl", 29535)


def test_two_ranks_small_batch_grouped_pipelined_schedule():
    """bs=2 per rank: the ranks run the grouped launches and the pipelined step (engine._pipelined_step: per-pair discriminator exchanges
    inside the task graph, generator ranges behind the milestone events of the grouped last backward pass)."""
    _run("gloo", 29537, per_rank=2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has fewer than two)")
def test_two_ranks_over_rccl_small_batch():
    _run("nccl", 29539, per_rank=2)


def test_two_ranks_across_the_identity_cutoff():
    """after the identity cut-off (train.py:314-315; 98 % of a default run) the merged-forward schedule ends, and five free-running
    persistent trunk passes no longer fit a data-parallel rank's compute units.  The ranks now order the generator phase's grouped forwards
    behind the discriminator phase's (engine._serial_fwd: one grouped persistent pass in flight, as on the merged schedule).  One sample per
    rank, four iterations, the cut-off after the second: merged forwards first, serialised separate passes afterwards -- bit-identical
    parameters on all ranks, rank-mean losses == one process on the concatenated minibatches with the same schedule."""
    out = _run("gloo", 29545, per_rank=1, MCVC_TEST_DDP_ITERS="4", MCVC_TEST_DDP_STOP_ID="2")
    assert "(True, False, 5.0), (True, False, 5.0), (False, True, 0.0), (False, True, 0.0)" in out, out[-1500:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has fewer than two)")
def test_two_ranks_over_rccl_across_the_identity_cutoff():
    _run("nccl", 29547, per_rank=1, MCVC_TEST_DDP_ITERS="4", MCVC_TEST_DDP_STOP_ID="2")
