"""GPU: the data-parallel engine with two real processes.  A dev box has ONE GPU, so both ranks share cuda:0 and exchange
over gloo (RCCL refuses duplicate devices); what is under test is the engine's choreography -- flat-buffer all-reduce,
1/R folded into Adam, the deferred discriminator update on the communication stream -- not the transport."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_match_one_process_and_each_other():
    env = dict(os.environ, MCVC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(HERE, "ddp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "parameter spread across ranks 0.000e+00" in r.stdout
