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


def batch_of(seeds):
    parts = [sample(s) for s in seeds]
    return [torch.cat([p[k] for p in parts], 0).contiguous() for k in range(4)]


def kink_free_seeds(eng, n, seed0, margin=3e-5):
    """|a-b| has a kinked gradient: an L1 residual within rounding of zero may legitimately take either sign in two
    differently tiled runs (bs=8 per rank vs bs=16 in one process), and ONE flipped sign is a 1e-3-level gradient error.
    Pick samples whose cycle / identity residuals all stay `margin` away from zero (every op is per-sample, so a batch of
    such samples is kink-free); probed with the engine itself at bs=1."""
    keep, s = [], seed0
    B0 = eng.B
    eng._use(1)
    while len(keep) < n:
        b = sample(s)
        for dst, src in zip(eng.static_in, b):
            dst.copy_(src)
        eng._run_phase("G")
        res = [(eng.mel["cycle_A"] - b[0]).abs().min(), (eng.mel["cycle_B"] - b[2]).abs().min(),
               (eng.out_B2A[1:] - b[0]).abs().min(), (eng.out_A2B[1:] - b[2]).abs().min()]      # identity halves
        if float(min(res)) > margin:
            keep.append(s)
        s += 1
        assert s < seed0 + 20 * n + 40, "no kink-free samples"
    eng._use(B0)
    return keep


N_IT = int(os.environ.get("MCVC_TEST_DDP_ITERS", "3"))
# identity cut-off of part (2)'s schedule (train.py:314-315; global_step advances by world x batch per iteration): with a small value the
# later iterations run the regime after the cut-off -- on data-parallel ranks the separate-pass pipelined schedule with its forwards ordered
# one behind the other (engine._serial_fwd)
STOP_ID = float(os.environ.get("MCVC_TEST_DDP_STOP_ID", "1e9"))
PER_RANK = int(os.environ.get("MCVC_TEST_DDP_PER_RANK", "8"))          # BASELINE configs[3]: bs=8 per GPU (2: the grouped + pipelined schedule)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=True)      # a mismatched collective would otherwise hang the GPU box
    # MCVC_TEST_DDP_BACKEND=nccl: RCCL with one GPU per rank (needs >= 2 GPUs); default gloo with both ranks on cuda:0 (a dev box has one)
    backend = os.environ.get("MCVC_TEST_DDP_BACKEND", "gloo")
    os.environ["MCVC_DIST_BACKEND"] = backend
    rank, world, local_rank = init_from_env()
    assert world == 2 and dist.get_backend() == backend
    dev = local_rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    cdev = torch.device("cuda", dev) if backend == "nccl" else torch.device("cpu")      # where small collectives' tensors live
    if rank == 0:
        print("backend %s, ranks on %s" % (backend, "one GPU each" if backend == "nccl" else "cuda:0 (shared)"), flush=True)
    # ---- (1) gradient equivalence of the generator phase at bs=8 per rank
    eng = TrainEngine(nets_for(700 + 10 * 0), PER_RANK, 64, schedule=StepSchedule(batch_size=PER_RANK, n_samples=64, world_size=world),
                      reducer=FlatGradReducer())
    assert eng.defer_d_update
    seeds = kink_free_seeds(eng, PER_RANK, 4000 + 1000 * rank)
    all_seeds = [None, None]
    dist.all_gather_object(all_seeds, seeds)
    mine = batch_of(seeds)
    for dst, src in zip(eng.static_in, mine):
        dst.copy_(src)
    eng._run_phase("G")
    eng.reducer.reduce_(eng.g_group.grad)
    avg = (eng.g_group.grad * eng.reducer.grad_scale).double().cpu()
    ok = True
    if rank == 0:
        both = batch_of(all_seeds[0] + all_seeds[1])
        solo = FlatGradReducer()
        solo.world, solo.active = 1, False      # the single-process reference must not issue collectives
        ref = TrainEngine(nets_for(700), 2 * PER_RANK, 64, schedule=StepSchedule(batch_size=2 * PER_RANK, n_samples=64), reducer=solo)
        for dst, src in zip(ref.static_in, both):
            dst.copy_(src)
        ref._run_phase("G")
        g2 = ref.g_group.grad.double().cpu()
        err = float((avg - g2).norm() / g2.norm())
        print("ddp grad (2 ranks x bs=%d) vs single-process bs=%d grad rel err %.3e" % (PER_RANK, 2 * PER_RANK, err), flush=True)
        ok = ok and err < 1e-4
        del ref
    # ---- (2) ranks stay identical through full iterations (deferred D update, async all-reduce)
    del eng
    eng2 = TrainEngine(nets_for(800), PER_RANK, 64, schedule=StepSchedule(batch_size=PER_RANK, n_samples=64, world_size=world, stop_identity_after=STOP_ID),
                       reducer=FlatGradReducer())
    mean_losses, mine_losses, sched_seen = [], [], []
    for it in range(N_IT):
        sched_seen.append((bool(eng2._use_merged()), bool(eng2._serial_fwd()), float(eng2.sched.identity_loss_lambda)))
        eng2.step(*batch_of(range(100 + 2 * PER_RANK * it + PER_RANK * rank, 100 + 2 * PER_RANK * it + PER_RANK * rank + PER_RANK)))
        # pipelined schedule (small batch): the losses of the last COMPLETE iteration, one step behind -- reading the current ones every
        # iteration would complete the pending discriminator phase and never exercise the pipelined graph
        lo = eng2.losses(lagged=True)
        if lo is not None and (eng2._pending_D is None or it > 0):
            mine_losses.append(lo)
    if eng2._pending_D is not None:
        mine_losses.append(eng2.losses())
    eng2.flush()
    assert len(mine_losses) == N_IT
    if rank == 0:
        print("pipelined schedule: %s" % bool(eng2._use_pipeline()), flush=True)
        print("per iteration (merged forwards, serialised forwards, identity lambda): %s" % sched_seen, flush=True)
    for lo in mine_losses:
        assert np.isfinite(lo["g_loss"]) and np.isfinite(lo["d_loss"])
        t = torch.tensor([lo["g_loss"], lo["d_loss"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        mean_losses.append((t / world).tolist())
    # ---- (3) the same three iterations in ONE process on the concatenated minibatches: the losses (means over the global batch) must
    # agree -- exactly the arithmetic before the first update, and closely after two (Adam's first steps amplify rounding)
    if rank == 0:
        solo = FlatGradReducer()
        solo.world, solo.active = 1, False
        ref = TrainEngine(nets_for(800), 2 * PER_RANK, 64, schedule=StepSchedule(batch_size=2 * PER_RANK, n_samples=64, stop_identity_after=STOP_ID),
                          reducer=solo)
        for it in range(N_IT):
            ref.step(*batch_of(range(100 + 2 * PER_RANK * it, 100 + 2 * PER_RANK * it + 2 * PER_RANK)))
            lo = ref.losses()
            for a, b in zip(mean_losses[it], (lo["g_loss"], lo["d_loss"])):
                tol = 1e-4 if it == 0 else 5e-2
                if abs(a - b) > tol * abs(b):
                    print("iteration %d: 2-rank mean loss %.6f vs single-process %.6f" % (it, a, b), flush=True)
                    ok = False
        print("2-rank losses vs single process: %s" % ("match" if ok else "MISMATCH"), flush=True)
        del ref
    for grp in (eng2.g_group, eng2.d_group):
        hi, lo_ = grp.flat.clone(), grp.flat.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        spread = float((hi - lo_).abs().max())
        if rank == 0:
            print("parameter spread across ranks %.3e" % spread, flush=True)
        ok = ok and spread == 0.0
    assert eng2.d_group.step == N_IT and eng2.g_group.step == N_IT
    flag = torch.tensor([1.0 if ok else 0.0], device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


if __name__ == "__main__":
    main()
