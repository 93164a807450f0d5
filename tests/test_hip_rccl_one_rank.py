"""GPU: RCCL executes on the one GPU a dev box has (SURVEY.md section 8e -- the reference has no distributed code).

The two-process tests of test_hip_ddp.py need gloo here (RCCL refuses two ranks on one device), so until a multi-GPU box runs the `nccl`
twins no RCCL kernel had ever run beside this engine.  A ONE-rank `nccl` group with the exchange forced on (FlatGradReducer(force=True))
closes what one GPU can close: library load under HSA_ENABLE_IPC_MODE_LEGACY=0, the communication stream's ordering against the backward
pass's milestone events with real RCCL kernels in it, and the residency budget of the persistent trunk kernels with RCCL's workgroups on
the chip (the "+1 pass" of engine._set_residency)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_rank_rccl_exchange_beside_the_persistent_trunk_is_bit_identical_and_fault_free():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_one_rank_worker.py")], env=env, capture_output=True, text=True, timeout=400)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(lines[-1][7:])
    print(json.dumps(res))
    assert res["backend"] == "nccl" and res["bit_identical"]
    for k in ("no_comm", "rccl"):
        info = res[k]
        assert info["faults"] == 0 and not info["trunk_fallback"]
        assert info["trunk_persistent"] & 1, "the persistent trunk kernels must be ON beside RCCL (residency 2 + 1 passes)"
        # merged forwards before the identity cut-off, serialised separate passes after it: the data-parallel schedule
        assert [tuple(s) for s in info["schedule"]] == [(True, False, 5.0), (True, False, 5.0), (False, True, 0.0), (False, True, 0.0)]
        assert info["resid"][0] == 3
    assert res["rccl"]["comm_waits_per_step"] > 0                     # the compute streams really waited on the communication stream
    assert all(abs(g) < 1e6 and abs(d) < 1e6 for g, d in res["losses"])
