"""GPU: grouped ("twin") launches -- two networks of the same architecture in one grid (csrc/twin.h, include/mcvc.h mcvc_twin_*).

The reference runs G_A2B / G_B2A and the discriminator pairs one after the other (train.py:203-216, 255-273); the four-lane schedule ran
them on separate streams; the grouped schedule launches every kernel of a pair of passes once with gridDim.z = 2.  Every kernel computes
exactly what it computes alone, so in bit-reproducible mode the grouped step must equal the four-lane step BIT FOR BIT."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mcvc_oracle as orc  # noqa: E402  (parameter filler only)
from mask_cyclegan_vc import _hip  # noqa: E402
from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream  # noqa: E402
from mask_cyclegan_vc.engine import D_NAMES, G_NAMES, TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402


def _nets(seed0):
    nets = {}
    for i, n in enumerate(orc.NET_ORDER):
        m = Generator() if i < 2 else Discriminator()
        m.load_state_dict(orc.filler_params("G" if i < 2 else "D", seed0 + i), strict=True)
        nets[n] = m.cuda()
    return nets


def _batch(B, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(2):
        out.append(torch.from_numpy(rs.randn(B, 80, 64).astype(np.float32)).cuda())
        out.append(torch.from_numpy(orc.fif_mask(rs, B, 80, 64, 25)).cuda())
    return out


@pytest.fixture
def deterministic_mode():
    L = lib()
    was = L.mcvc_set_deterministic(1)
    yield
    L.mcvc_set_deterministic(was)


def _steps(grouped, B, n_it=2):
    nets = _nets(610)
    eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=4 * B))
    eng.grouped = grouped
    eng.merged = False                     # (the merged forwards have their own test: other tile shapes, agreement to rounding)
    losses = []
    for it in range(n_it):
        eng.step(*_batch(B, 70 + it))
        losses.append(eng.losses())
    eng.flush()
    eng.check_faults()
    return losses, {n: [p.detach().clone() for p in nets[n].parameters()] for n in G_NAMES + D_NAMES}


@pytest.mark.parametrize("B", [1, 2, 4])
def test_grouped_step_is_bit_identical_to_the_four_lane_step(deterministic_mode, B):
    (l0, p0), (l1, p1) = _steps(False, B), _steps(True, B)
    assert l0 == l1, (l0, l1)
    for n in p0:
        for i, (a, b) in enumerate(zip(p0[n], p1[n])):
            assert torch.equal(a, b), (n, i)


@pytest.mark.parametrize("B", [1, 4])
def test_pipelined_steps_equal_back_to_back_phases(deterministic_mode, B):
    """The pipelined step issues iteration t's discriminator phase together with iteration t+1's generator phase (engine._pipelined_step).
    Every quantity is computed from the weights and inputs the reference uses, so K pipelined steps + flush == K steps with the two phases
    back to back: losses of every iteration and all parameters bit-equal (deterministic mode), Adam step counts equal."""
    def run(pipelined, K=4):
        nets = _nets(640)
        eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=4 * B, decay_after=2 * B, stop_identity_after=3 * B, num_epochs=3))
        eng.pipelined = pipelined
        eng.merged = False                 # (separate D-phase generator forwards: the same kernels on the same shapes as the phases back to back)
        seen = []
        for it in range(K):
            eng.step(*_batch(B, 90 + it))
            if pipelined:
                assert eng._pending_D is not None
                lo = eng.losses(lagged=True)              # iteration it-1 (None at the first step)
                assert (lo is None) == (it == 0)
                if lo is not None:
                    seen.append((lo["g_loss"], lo["d_loss"]))
            else:
                lo = eng.losses()
                seen.append((lo["g_loss"], lo["d_loss"]))
        if pipelined:
            lo = eng.losses()                             # completes the last iteration
            seen.append((lo["g_loss"], lo["d_loss"]))
            assert eng._pending_D is None
        eng.flush()
        eng.check_faults()
        return seen, {n: [p.detach().clone() for p in nets[n].parameters()] for n in G_NAMES + D_NAMES}, (eng.g_group.step, eng.d_group.step), \
            (eng.sched.g_opt_lr, eng.sched.d_opt_lr, eng.sched.global_step)
    (l0, p0, s0, h0), (l1, p1, s1, h1) = run(False), run(True)
    assert s0 == s1 == (4, 4) and h0 == h1
    assert l0 == l1, (l0, l1)
    for n in p0:
        for i, (a, b) in enumerate(zip(p0[n], p1[n])):
            assert torch.equal(a, b), (n, i)


def test_loss_readback_two_steps_behind_returns_the_same_losses_without_waiting(deterministic_mode):
    """``losses(lagged=2)`` -- what the training loop and bench.py read -- is the iteration issued two step()s earlier: the values
    ``lagged=True`` returned one step() before, None until there is one, and after a synchronous read (``losses()``) the ring starts
    over."""
    B = 1
    eng = TrainEngine(_nets(640), B, 64, schedule=StepSchedule(batch_size=B, n_samples=4 * B, num_epochs=3))
    newest, two_behind = [], []
    for it in range(5):
        eng.step(*_batch(B, 90 + it))
        lo2 = eng.losses(lagged=2)                 # first: must not depend on the lag-1 read having waited
        lo1 = eng.losses(lagged=True)
        assert (lo1 is None) == (it == 0) and (lo2 is None) == (it <= 1)
        newest.append(lo1)
        two_behind.append(lo2)
    assert two_behind[2:] == newest[1:4]
    last = eng.losses()                            # flush: completes iteration 4
    assert eng._pending_D is None and np.isfinite(last["g_loss"]) and last != newest[4]
    eng.step(*_batch(B, 99))
    assert eng.losses(lagged=True) is None and eng.losses(lagged=2) is None
    eng.flush()
    eng.check_faults()


def test_merged_forwards_equal_the_separate_passes(deterministic_mode):
    """The default bs=1 step batches iteration t's discriminator-phase generator forwards INTO iteration t+1's generator-phase passes
    (engine._merged_step: three / two samples per pass, backward over the first two / one).  Every op of the generator is per sample, so all
    losses of every iteration and all parameters agree with the separate passes of _pipelined_step to rounding (a three-sample pass picks
    other tile shapes and K splits than a two-sample one: not bit-equal); both are bit-reproducible run to run; Adam step counts, learning
    rates and the identity cut-off are those of the reference schedule; and the merged step issues fewer launches."""
    B, K = 1, 5

    def run(merged):
        nets = _nets(640)
        eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=4 * B, decay_after=2 * B, stop_identity_after=3 * B, num_epochs=3))
        eng.merged = merged
        eng._use(B)                        # (re-bind: the residency bound of the persistent trunk kernels follows the schedule)
        assert eng._use_merged() == merged
        seen = []
        for it in range(K):
            eng.step(*_batch(B, 90 + it))
            lo = eng.losses(lagged=True)
            assert (lo is None) == (it == 0)
            if lo is not None:
                seen.append(lo)
        seen.append(eng.losses())
        eng.flush()
        eng.check_faults()
        return seen, {n: [p.detach().clone() for p in nets[n].parameters()] for n in G_NAMES + D_NAMES}, (eng.g_group.step, eng.d_group.step), \
            (eng.sched.g_opt_lr, eng.sched.d_opt_lr, eng.sched.global_step)
    (l0, p0, s0, h0), (l1, p1, s1, h1), (l2, p2, s2, h2) = run(False), run(True), run(True)
    assert s0 == s1 == (K, K) and h0 == h1
    assert l1 == l2 and all(torch.equal(a, b) for n in p1 for a, b in zip(p1[n], p2[n]))          # bit-reproducible
    for lx, px in ((l1, p1),):
        for k in l0[0]:                    # first iteration: the same forward arithmetic up to tile shapes
            assert abs(l0[0][k] - lx[0][k]) <= 2e-6 * abs(l0[0][k]) + 1e-8, (k, l0[0], lx[0])
        for it, (a, b) in enumerate(zip(l0, lx)):
            for k in a:
                assert abs(a[k] - b[k]) < 1e-3 * abs(a[k]) + 1e-7, (it, k, a, b)
        for n in p0:
            for a, b in zip(p0[n], px[n]):
                assert float((a - b).norm()) <= 0.1 * K * 2e-4 * float(a.numel()) ** 0.5 + 1e-7, n      # 10 % of "every element moved by lr" per step


def test_grouped_step_default_mode_close_and_halves_the_launches():
    """Default (atomics) mode: the two schedules agree to rounding; and the point of the exercise -- a traced grouped step issues
    about half the kernel launches of the four-lane step."""
    import ctypes
    L = lib()
    (l0, p0), (l1, p1) = _steps(False, 1), _steps(True, 1)
    for a, b in zip(l0, l1):
        for k in ("g_loss", "d_loss"):
            assert abs(a[k] - b[k]) < 2e-3 * abs(a[k]), (k, a, b)
    counts = {}
    for grouped in (False, True):
        eng = TrainEngine(_nets(610), 1, 64, schedule=StepSchedule(batch_size=1, n_samples=4))
        eng.grouped = grouped
        eng.step(*_batch(1, 5))
        eng.flush()                        # (a pending pipelined discriminator phase would otherwise be counted with the traced step)
        torch.cuda.synchronize()
        eng.concurrent = False
        eng.aux_wgrad = False
        nk = L.mcvc_trace_kinds()
        buf = (ctypes.c_double * (4 * nk))()
        L.mcvc_trace_enable(1)
        eng.step(*_batch(1, 6))
        L.mcvc_trace_collect(buf)
        L.mcvc_trace_enable(0)
        counts[grouped] = (int(sum(buf[4 * k] for k in range(nk))), sum(buf[4 * k + 2] for k in range(nk)))
    print("launches per step: four-lane %d, grouped %d" % (counts[False][0], counts[True][0]))
    assert counts[True][0] <= 0.6 * counts[False][0]
    assert abs(counts[True][1] - counts[False][1]) < 1e-6 * counts[False][1]        # same executed FLOPs, counted once per network


def test_twin_bracket_on_the_c_abi_and_its_mismatch_guard():
    """Two generator forwards through mcvc_twin_begin / switch / end equal the two single calls bit for bit; two sequences that do not
    issue the same launches (different batch) are refused with an error instead of being paired up wrongly."""
    L = lib()
    torch.manual_seed(0)
    gens = [Generator().cuda(), Generator().cuda()]
    B, T = 2, 64
    xs = [torch.randn(B, 80, T, device="cuda") for _ in range(2)]
    ms = [torch.ones(B, 80, T, device="cuda") for _ in range(2)]
    ms[0][:, :, 10:20] = 0.0
    tabs = [ptr_table(list(g.parameters())) for g in gens]
    packed = [g.packed_weights() for g in gens]

    def fwd(i, out, stash, scratch, nb=B):
        check(L.mcvc_gen_forward(tabs[i], ptr(packed[i]), ptr(xs[i]), ptr(ms[i]), ptr(out), ptr(stash), ptr(scratch), scratch.numel(), nb, T,
                                 stream()), "gen_forward")
    mk = lambda: (torch.zeros(B, 80, T, device="cuda"), torch.zeros(L.mcvc_gen_stash_floats(B, T), device="cuda"),      # noqa: E731
                  torch.zeros(L.mcvc_gen_scratch_floats(B, T), device="cuda"))
    single = [mk(), mk()]
    for i in range(2):
        fwd(i, *single[i])
    twin = [mk(), mk()]
    with _hip.twin() as tw:
        fwd(0, *twin[0])
        tw.switch()
        fwd(1, *twin[1])
    n_single = L.mcvc_twin_launches()
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(twin[i][0], single[i][0]) and torch.equal(twin[i][1], single[i][1]), i
    assert n_single > 20
    assert not torch.equal(single[0][0], single[1][0])
    # mismatch: the second sequence runs a different batch size -> different grids
    assert L.mcvc_twin_begin() == 0
    fwd(0, *twin[0])
    assert L.mcvc_twin_switch() == 0
    L.mcvc_gen_forward(tabs[1], ptr(packed[1]), ptr(xs[1]), ptr(ms[1]), ptr(twin[1][0]), ptr(twin[1][1]), ptr(twin[1][2]), twin[1][2].numel(), 1, T,
                       stream())
    assert L.mcvc_twin_end() != 0
    assert L.mcvc_twin_end() != 0 and L.mcvc_twin_begin() == 0 and L.mcvc_twin_switch() == 0 and L.mcvc_twin_end() == 0      # bracket state is clean again
    torch.cuda.synchronize()
