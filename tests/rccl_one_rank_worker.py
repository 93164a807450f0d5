"""Worker for tests/test_hip_rccl_one_rank.py: ONE rank, backend nccl (= RCCL), on the single GPU of a dev box.

A 1-rank all-reduce is the identity, but it is a real RCCL collective: the library loads (HSA_ENABLE_IPC_MODE_LEGACY=0), its kernels run on
the engine's communication stream behind the milestone events of the grouped last backward pass, beside the persistent trunk kernels whose
in-kernel hand-off needs all their workgroups resident (csrc/trunk.h; the engine budgets one pass's worth of compute units for RCCL).
The data-parallel schedule (merged forwards before the identity cut-off, serialised forwards after it, deferred discriminator update) with
the exchange FORCED on must therefore equal the same schedule with a reducer that issues nothing -- bit for bit in deterministic mode --
and report no persistent-kernel fault.  Prints one JSON line with the exposed communication time and the host enqueue time per step."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("maskcyclegan-vc_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import mcvc_oracle as orc  # noqa: E402  (parameter filler only)
from mask_cyclegan_vc import _hip  # noqa: E402
from mask_cyclegan_vc.engine import TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.parallel import FlatGradReducer  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402

N_IT = 4


class NoCommReducer(FlatGradReducer):
    """The rank schedule with nothing exchanged: the comparison run."""

    def __init__(self):
        super().__init__()
        self.active = True

    def reduce_(self, flat):
        return flat

    def reduce_async_(self, flat):
        return flat

    def reduce_range_after_(self, flat, lo, hi, event=None):
        return

    def wait(self, device=None):
        return

    def broadcast_(self, flat, src=0):
        return flat


def nets_for(seed0):
    nets = {}
    for i, n in enumerate(orc.NET_ORDER):
        m = Generator() if i < 2 else Discriminator()
        m.load_state_dict(orc.filler_params("G" if i < 2 else "D", seed0 + i), strict=True)
        nets[n] = m.cuda()
    return nets


def batch(seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(2):
        out.append(torch.from_numpy(rs.randn(1, 80, 64).astype(np.float32)).cuda())
        out.append(torch.from_numpy(orc.fif_mask(rs, 1, 80, 64, 25)).cuda())
    return out


def run(reducer, timed):
    eng = TrainEngine(nets_for(900), 1, 64, schedule=StepSchedule(batch_size=1, n_samples=8, stop_identity_after=1), reducer=reducer)
    assert eng._use_pipeline() and eng.defer_d_update and eng.overlap_g_reduce
    reducer.time_waits = timed
    seen, losses, host = [], [], 0.0
    for it in range(N_IT):
        seen.append((bool(eng._use_merged()), bool(eng._serial_fwd()), float(eng.sched.identity_loss_lambda)))
        b = batch(50 + it)
        t0 = time.perf_counter()
        eng.step(*b)
        host += time.perf_counter() - t0
        lo = eng.losses(lagged=True)
        if lo is not None:
            losses.append(lo)
    losses.append(eng.losses())
    eng.flush()
    faults = eng.check_faults(raise_on_fault=False)
    exposed, waits = reducer.exposed_ms() if timed else (0.0, 0)
    L = _hip.lib()
    pers = L.mcvc_gen_trunk_persistent(1, 64)
    params = torch.cat([eng.g_group.flat, eng.d_group.flat]).clone()
    info = dict(schedule=seen, faults=int(faults), trunk_persistent=int(pers), trunk_fallback=bool(eng.trunk_fallback), resid=list(eng._resid),
                exposed_comm_ms_per_step=exposed / N_IT, comm_waits_per_step=waits / N_IT, host_enqueue_ms_per_step=1e3 * host / N_IT)
    return params, [(lo["g_loss"], lo["d_loss"]) for lo in losses], info


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    L = _hip.lib()
    L.mcvc_set_deterministic(1)
    assert L.mcvc_set_trunk_persistent(1) in (0, 1, 2)
    # a real collective before anything else: RCCL initialises its communicator (and proves the library loads on this image)
    t = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t.sum()) == float(1 << 20)
    p0, l0, i0 = run(NoCommReducer(), timed=False)
    forced = FlatGradReducer(force=True)
    assert forced.active and forced.world == 1 and forced.grad_scale == 1.0
    p1, l1, i1 = run(forced, timed=True)
    ok = bool(torch.equal(p0, p1)) and l0 == l1
    out = dict(bit_identical=ok, backend=dist.get_backend(), losses=l1, no_comm=i0, rccl=i1)
    print("RESULT " + json.dumps(out), flush=True)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
