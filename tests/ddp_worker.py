"""Worker for tests/test_hip_ddp.py: two ranks (gloo, both on cuda:0 -- a dev box has one GPU) run the data-parallel
engine; rank 0 checks (1) averaged per-rank generator-phase gradients == the gradient of one process on the concatenated
batch, (2) after full iterations with the deferred discriminator update every rank holds identical parameters."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("maskcyclegan-vc_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import mcvc_oracle as orc  # noqa: E402  (parameter filler only)
from mask_cyclegan_vc.engine import D_NAMES, G_NAMES, TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.parallel import FlatGradReducer, init_from_env  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402


def nets_for(seed0):
    nets = {}
    for i, n in enumerate(orc.NET_ORDER):
        m = Generator() if i < 2 else Discriminator()
        m.load_state_dict(orc.filler_params("G" if i < 2 else "D", seed0 + i), strict=True)
        nets[n] = m.cuda()
    return nets


def sample(seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(2):
        real = torch.from_numpy(rs.randn(1, 80, 64).astype(np.float32))
        mask = torch.ones(1, 80, 64)
        size = int(rs.randint(1, 25)); start = int(rs.randint(0, 64 - size))
        mask[0, :, start:start + size] = 0.0
        out += [real.cuda(), mask.cuda()]
    return out


def main():
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)      # a mismatched collective would otherwise hang the GPU box
    os.environ["MCVC_DIST_BACKEND"] = "gloo"
    rank, world, _ = init_from_env()
    assert world == 2
    torch.cuda.set_device(0)
    # ---- (1) gradient equivalence of the generator phase
    eng = TrainEngine(nets_for(700 + 10 * 0), 1, 64, schedule=StepSchedule(batch_size=1, n_samples=8, world_size=world),
                      reducer=FlatGradReducer())
    assert eng.defer_d_update
    mine = sample(40 + rank)
    for dst, src in zip(eng.static_in, mine):
        dst.copy_(src)
    eng._run_phase("G")
    eng.reducer.reduce_(eng.g_group.grad)
    avg = (eng.g_group.grad * eng.reducer.grad_scale).double().cpu()
    ok = True
    if rank == 0:
        both = [torch.cat([a, b]) for a, b in zip(sample(40), sample(41))]
        solo = FlatGradReducer()
        solo.world = 1                  # the single-process reference must not issue collectives
        ref = TrainEngine(nets_for(700), 2, 64, schedule=StepSchedule(batch_size=2, n_samples=8), reducer=solo)
        for dst, src in zip(ref.static_in, both):
            dst.copy_(src)
        ref._run_phase("G")
        g2 = ref.g_group.grad.double().cpu()
        err = float((avg - g2).norm() / g2.norm())
        print("ddp grad vs batch-2 grad rel err %.3e" % err, flush=True)
        ok = ok and err < 2e-2          # (L1 kinks: one flipped sign is 0.5 % -- see test_hip_engine.py)
    # ---- (2) ranks stay identical through full iterations (deferred D update, async all-reduce)
    eng2 = TrainEngine(nets_for(800), 1, 64, schedule=StepSchedule(batch_size=1, n_samples=8, world_size=world), reducer=FlatGradReducer())
    for it in range(3):
        eng2.step(*sample(100 + 2 * it + rank))
        lo = eng2.losses()
        assert np.isfinite(lo["g_loss"]) and np.isfinite(lo["d_loss"])
    eng2.flush()
    for grp in (eng2.g_group, eng2.d_group):
        hi, lo_ = grp.flat.clone(), grp.flat.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        spread = float((hi - lo_).abs().max())
        if rank == 0:
            print("parameter spread across ranks %.3e" % spread, flush=True)
        ok = ok and spread == 0.0
    assert eng2.d_group.step == 3 and eng2.g_group.step == 3
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


if __name__ == "__main__":
    main()
