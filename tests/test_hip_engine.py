"""GPU: the hand-scheduled training step (engine.py) against the reference's UNMODIFIED train() loop,
replayed from golden fixtures (losses + per-parameter norms after every iteration), and against the
CPU oracle on full tensors.  Bar 1e-3 relative."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mcvc_oracle as orc  # noqa: E402
from mask_cyclegan_vc.engine import D_NAMES, G_NAMES, TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402


@pytest.fixture
def deterministic_mode():
    from mask_cyclegan_vc import _hip
    L = _hip.lib()
    was = L.mcvc_set_deterministic(1)
    yield
    L.mcvc_set_deterministic(was)


def _nets(seeds):
    nets = {}
    for i, (n, s) in enumerate(zip(orc.NET_ORDER, seeds)):
        m = Generator() if i < 2 else Discriminator()
        m.load_state_dict(orc.filler_params("G" if i < 2 else "D", s), strict=True)
        nets[n] = m.cuda()
    return nets


def _zero_grad_bias_names(golden_dir):
    """Conv biases directly in front of an InstanceNorm have a mathematically zero gradient; the reference's
    autograd leaves ~1e-9 noise there which Adam (eps 1e-8) turns into a random walk of ~lr per step.  Those
    parameters cannot influence any output (the norm removes them) and are excluded from multi-step parameter
    parity (SURVEY.md section 7 "hard parts"); the HIP path leaves them exactly unchanged."""
    norms = json.load(open(os.path.join(golden_dir, "grad_norms.json")))
    return {k.split(":", 1)[1] for k, v in norms.items() if v is not None and v < 1e-6}


def _run_against_golden(golden_dir, tag, n_it):
    skip = _zero_grad_bias_names(golden_dir)
    assert 20 < len(skip) < 40 and all(k.endswith(".bias") for k in skip)
    js = json.load(open(os.path.join(golden_dir, "step_%s.json" % tag)))
    bt = np.load(os.path.join(golden_dir, "step_%s_batches.npz" % tag))
    cfg = js["config"]
    nets = _nets(cfg["filler_seeds"])
    bs = cfg["batch_size"]
    sched = StepSchedule(generator_lr=cfg["g_lr"], discriminator_lr=cfg["d_lr"], num_epochs=cfg["num_epochs"], n_samples=cfg["n_utt"],
                         batch_size=bs, decay_after=cfg["decay_after"], stop_identity_after=cfg["stop_identity_after"])
    eng = TrainEngine(nets, bs, 64, schedule=sched)
    for it in range(n_it):
        batch = [torch.from_numpy(bt["it%d_%s" % (it, k)]).cuda() for k in ("real_A", "mask_A", "real_B", "mask_B")]
        # the fixture snapshots optimizer lr inside logger.end_iter(), i.e. BEFORE this iteration's lr adjustment (train.py:304-311)
        lr_before = (sched.g_opt_lr, sched.d_opt_lr)
        eng.step(*batch)
        lo = eng.losses()
        eng.flush()                          # the discriminators' Adam step is deferred to the next use: land it before reading parameters
        ref = js["losses"][it]
        assert abs(lo["g_loss"] - ref["g_loss"]) < 1e-3 * abs(ref["g_loss"]), (it, lo, ref)
        assert abs(lo["d_loss"] - ref["d_loss"]) < 1e-3 * abs(ref["d_loss"]), (it, lo, ref)
        tr = js["trace"][it]
        assert sched.global_step == tr["global_step"]
        assert abs(lr_before[0] - tr["g_opt_lr"]) < 1e-12 and abs(lr_before[1] - tr["d_opt_lr"]) < 1e-12
        for name in orc.NET_ORDER:
            for (pn, p), rn in zip(nets[name].named_parameters(), tr["norms"][name]):
                if pn in skip:
                    continue
                mine = float(p.detach().double().norm())
                # single-element parameters (the discriminators' 1-channel output bias): their gradient is a sum of
                # opposite-signed real / fake terms, so rounding is amplified by cancellation and then by Adam's
                # normalisation; observed run-to-run spread 2e-3 of a value of 5e-3 (one lr step would be 2e-2 of it)
                tol = 1e-2 if p.numel() == 1 else 1e-3
                assert abs(mine - rn) <= tol * max(rn, 1e-3), (it, name, pn)
    return eng, js


def test_three_iterations_match_unmodified_reference_train(golden_dir, deterministic_mode):
    eng, js = _run_against_golden(golden_dir, "plain", 3)
    sd = eng.optimizer_state_dict("D")
    assert sorted(sd["state"].keys()) == js["adam_state_keys_D"]
    assert sorted(eng.optimizer_state_dict("G")["state"].keys()) == js["adam_state_keys_G"]


def test_lr_decay_bug_and_identity_cutoff_match_reference(golden_dir, deterministic_mode):
    eng, js = _run_against_golden(golden_dir, "decay", 2)
    fin = js["final"]
    assert eng.sched.identity_loss_lambda == fin["identity_loss_lambda"] == 0
    assert abs(eng.sched.g_opt_lr - fin["g_opt_lr"]) < 1e-12
    assert abs(eng.sched.generator_lr - fin["generator_lr_attr"]) < 1e-12


def test_four_iterations_past_the_identity_cutoff_pipelined_and_unflushed(golden_dir, deterministic_mode):
    """The regime a canonical run spends > 97 % of its life in (train.py:314-315: identity_loss_lambda = 0 from then on; :207-210 keeps
    computing the identity forwards, which the engine drops as dead work -- ``ident_dead`` in engine._g_parts), pinned to the REFERENCE:
    the ``cutoff`` fixture is four iterations of the unmodified train() at bs=2 whose iterations 2 and 3 run with lambda_id == 0.
    Replayed through the DEFAULT schedule -- grouped launches, iteration t's discriminator phase pipelined beside iteration t+1's
    generator phase -- with NO flush() between the steps: losses are read the way the training loop reads them (``lagged=True``), the
    parameters after the single flush at the end.  Gates: losses 1e-3; per parameter tensor the norm (fixture trace) 1e-3, the fixture's
    element samples and the full tensor against the oracle (itself pinned to the same fixture in test_oracle_golden.py): see below."""
    skip = _zero_grad_bias_names(golden_dir)
    js = json.load(open(os.path.join(golden_dir, "step_cutoff.json")))
    bt = np.load(os.path.join(golden_dir, "step_cutoff_batches.npz"))
    cfg = js["config"]
    bs, n_it = cfg["batch_size"], 4
    nets = _nets(cfg["filler_seeds"])
    sched = StepSchedule(generator_lr=cfg["g_lr"], discriminator_lr=cfg["d_lr"], num_epochs=cfg["num_epochs"], n_samples=cfg["n_utt"],
                         batch_size=bs, decay_after=cfg["decay_after"], stop_identity_after=cfg["stop_identity_after"])
    eng = TrainEngine(nets, bs, 64, schedule=sched)
    assert eng._use_pipeline(), "the default schedule at this batch size is the pipelined one"
    onets = {n: orc.filler_params("G" if i < 2 else "D", s) for i, (n, s) in enumerate(zip(orc.NET_ORDER, cfg["filler_seeds"]))}
    so = orc.StepOracle(onets, skip_wasted=True)
    got, lam = [], []
    for it in range(n_it):
        batch = [torch.from_numpy(bt["it%d_%s" % (it, k)]) for k in ("real_A", "mask_A", "real_B", "mask_B")]
        lam.append(sched.identity_loss_lambda)
        eng.step(*[b.cuda() for b in batch])
        assert eng._pending_D is not None                       # (iteration `it`'s discriminator phase has NOT run yet)
        lo = eng.losses(lagged=True)                            # iteration it - 1, complete; never waits for work in flight
        assert (lo is None) == (it == 0)
        if lo is not None:
            got.append(lo)
        so.identity_lambda = float(lam[-1])
        so.step(*batch)                                         # the oracle, same bookkeeping (no lr decay in this fixture)
    assert lam == [t["identity_lambda_before_check"] for t in js["trace"]] == [5, 5, 0, 0]
    eng.flush()
    got.append(eng.losses())
    assert sched.identity_loss_lambda == js["final"]["identity_loss_lambda"] == 0 and sched.global_step == js["trace"][-1]["global_step"]
    for it, (lo, ref) in enumerate(zip(got, js["losses"])):
        assert abs(lo["g_loss"] - ref["g_loss"]) < 1e-3 * abs(ref["g_loss"]), (it, lo, ref)
        assert abs(lo["d_loss"] - ref["d_loss"]) < 1e-3 * abs(ref["d_loss"]), (it, lo, ref)
    assert got[2]["identity_loss"] == 0.0 and got[3]["identity_loss"] == 0.0 and got[1]["identity_loss"] > 0.0
    # Per-tensor distance after FOUR Adam steps, in DETERMINISTIC mode: one number per binary.  What bounds it is stated against an fp64
    # anchor in tests/test_hip_parity_fp64.py (HIP and the reference's fp32 arithmetic each sit ~1e-3 at worst from the float64 result:
    # Adam's first steps turn rounding-level gradients into +-lr parameter differences); here the HIP step is compared with the reference's
    # fp32 result directly:
    #   e_s  against the fixture's element samples -- fixed numbers on both sides, so the gate is the measured worst x 1.25 and the count of
    #        tensors beyond 1e-3 is gated on THIS distance (host-independent);
    #   e_f  against the oracle's full tensors.  The oracle runs on this host's CPU, and its result is a property of that CPU: the same oracle
    #        on an AVX-512 Xeon / 8 threads sits 2.0e-3 from the fixture on upSample1 / downSample2 with 15 of 232 tensors beyond 1e-3
    #        (tests/test_oracle_golden.py::test_oracle_past_the_identity_cutoff), on the MI355X boxes' EPYC 9575F under 1e-3.  Its gate is the
    #        reference's asserted self-spread (7e-3 / 2, test_reference_arithmetic_spread_after_four_adam_steps), not a number tuned on one host.
    # r6 binary (two-level accumulation in the Winograd GEMMs): worst sample distance 1.383e-3 with 9 of 232 tensors beyond 1e-3 (gate 5 % = 11);
    # against the EPYC boxes' oracle worst full-tensor distance < 1e-3, 0 beyond (profiles/r06b_parity_host_independent.log).  (r5: 4.2e-3 - 4.6e-3, 63-65 tensors beyond 1e-3, gate 7e-3: profiles/r06_parity_probe_before.log has the why.)
    worst, n_t, n_over_s, n_over_f, bad, top = 0.0, 0, 0, 0, [], []
    for name in orc.NET_ORDER:
        for j, ((pn, p), rn) in enumerate(zip(nets[name].named_parameters(), js["trace"][-1]["norms"][name])):
            if pn in skip:
                continue
            mine = p.detach().cpu()
            tol = 1e-2 if p.numel() == 1 else 1e-3              # (the one-element output bias: see _run_against_golden)
            assert abs(float(mine.double().norm()) - rn) <= tol * max(rn, 1e-3), (name, pn)
            if p.numel() == 1:
                continue
            flat = mine.flatten()
            ref = bt["final_%s_%d" % (name, j)]
            e_s = float(np.linalg.norm(flat[torch.from_numpy(orc.sample_index(flat.numel()))].numpy().astype(np.float64) - ref)
                        / max(np.linalg.norm(ref.astype(np.float64)), 1e-30))
            e_f = float((mine.double() - onets[name][pn].double()).norm() / onets[name][pn].double().norm())
            worst = max(worst, e_s, e_f)
            top.append((max(e_s, e_f), name, pn, p.numel()))
            n_t += 1
            n_over_s += int(e_s > 1e-3)
            n_over_f += int(e_f > 1e-3)
            if not (e_s < 1.383e-3 * 1.25 and e_f < 7e-3 / 2):
                bad.append((name, pn, e_s, e_f))
    print("cutoff fixture, 4 un-flushed pipelined iterations: worst per-tensor rel-L2 %.3e; of %d tensors beyond 1e-3: %d (fixture samples), %d (this host's oracle)"
          % (worst, n_t, n_over_s, n_over_f))
    print("  largest: " + "; ".join("%s.%s (%d) %.2e" % (n, q, k, e) for e, n, q, k in sorted(top, reverse=True)[:6]))
    assert not bad, bad
    assert n_over_s <= 0.05 * n_t, (n_over_s, n_t)


def _kink_free_batch(onets, B, T=64):
    """|a-b| terms have a discontinuous gradient at a == b: an element that lands within rounding of the kink
    gets sign(+/-) from either side legitimately (observed: exactly one flipped element of 5120 => 2/sqrt(5120)
    relative error in that output-gradient, ~6e-3 in every upstream parameter gradient).  Pick seeded samples whose
    L1 residuals all stay away from zero so the comparison is well posed.  Every op of the step is per-sample
    (InstanceNorm has no batch statistics), so a batch of individually kink-free samples is kink-free -- which is what
    makes bs=8 / bs=32 feasible: a whole random batch of 32 has ~10 elements inside the margin."""
    picked = []
    so = orc.StepOracle(onets)
    for seed in range(5, 400):
        rs = np.random.RandomState(seed)
        mm = min(25, T)
        one = [torch.from_numpy(rs.randn(1, 80, T).astype(np.float32)), torch.from_numpy(orc.fif_mask(rs, 1, 80, T, mm)),
               torch.from_numpy(rs.randn(1, 80, T).astype(np.float32)), torch.from_numpy(orc.fif_mask(rs, 1, 80, T, mm))]
        with torch.no_grad():
            _, aux = so.losses_g(*one)
        res = [(aux["cycle_A"] - one[0]).abs().min(), (aux["cycle_B"] - one[2]).abs().min(),
               (aux["identity_A"] - one[0]).abs().min(), (aux["identity_B"] - one[2]).abs().min()]
        if float(min(res)) > 4e-5:      # GPU-vs-CPU forward differences are ~1e-5 absolute
            picked.append(one)
            if len(picked) == B:
                return [torch.cat([p[k] for p in picked], 0).contiguous() for k in range(4)]
    raise AssertionError("no kink-free batch found")


@pytest.mark.parametrize("B,T,lam_id", [(1, 64, 5.0), (2, 64, 5.0), (1, 32, 5.0), (3, 48, 5.0), (8, 64, 5.0), (32, 64, 5.0), (1, 64, 0.0), (8, 64, 0.0)])
def test_step_full_tensor_parity_vs_oracle(golden_dir, B, T, lam_id):
    """One iteration: every generator / discriminator parameter GRADIENT (full tensors) and the resulting Adam update
    vs the CPU oracle.  (8, 64) is the per-GPU shape of BASELINE configs[3], (32, 64) is configs[2]: the staged-GEMM trunk
    (more than 64 columns), the F(4x4,5x5) / F(4x4,3x3) Winograd schemes in chunks of samples, weight gradients over 64 images in the
    discriminator phase and Adam -- the full step with the L1 terms on.  ``lam_id`` = 0: the schedule after the identity cut-off
    (train.py:314-315), where the engine does not compute the identity passes at all."""
    seeds = [300 + i for i in range(6)]
    nets = _nets(seeds)
    onets = {n: orc.filler_params("G" if i < 2 else "D", s) for i, (n, s) in enumerate(zip(orc.NET_ORDER, seeds))}
    batch = _kink_free_batch(onets, B, T)
    so = orc.StepOracle(onets, skip_wasted=True, identity_lambda=lam_id)
    before = {n: {k: v.clone() for k, v in onets[n].items()} for n in onets}
    g_ref, d_ref, g_grads, d_grads = so.step(*batch, return_grads=True)
    eng = TrainEngine(nets, B, T, schedule=StepSchedule(batch_size=B, n_samples=4 * B, identity_loss_lambda=lam_id))
    # the iteration, phase by phase (what eng.step() does), so that the discriminator phase can start from the ORACLE's
    # updated generators: Adam's first step is lr*sign(g), i.e. rounding-level differences in near-zero generator gradients
    # become +-2e-4 parameter differences, and comparing discriminator gradients downstream of two such generator sets
    # measures that amplification, not the discriminator-phase kernels
    for dst, src in zip(eng.static_in, [b.cuda() for b in batch]):
        dst.copy_(src)
    eng._run_phase("G")
    g_grad_snapshot = {name: {k: p.grad.detach().clone() for k, p in nets[name].named_parameters()} for name in G_NAMES}
    eng.generator_update()
    g_after = {name: {k: p.detach().clone() for k, p in nets[name].named_parameters()} for name in G_NAMES}
    for name in G_NAMES:
        for k, p in nets[name].named_parameters():
            p.data.copy_(onets[name][k].to(p.device))
    eng._run_phase("D")
    eng.discriminator_update()
    eng.flush()
    eng.sched.end_iteration()
    lo = eng.losses()
    assert abs(lo["g_loss"] - g_ref) < 1e-4 * abs(g_ref) and abs(lo["d_loss"] - d_ref) < 1e-4 * abs(d_ref)
    gn, dn = orc.generator_param_names(), orc.discriminator_param_names()
    worst_g = worst_u = 0.0
    gi = iter(g_grads)
    for name in G_NAMES:
        mod = dict(nets[name].named_parameters())
        for k in gn:
            gr = next(gi)
            if float(gr.abs().max()) < 1e-6:          # zero-gradient bias class
                continue
            e = float((g_grad_snapshot[name][k].cpu().double() - gr.double()).norm() / gr.double().norm())
            worst_g = max(worst_g, e)
            assert e < 1e-3, (name, k, e)
            # Adam's first step is ~lr*sign(g): compare the update on the elements whose gradient is significant
            upd_ref = (onets[name][k] - before[name][k]).double()
            upd = (g_after[name][k].cpu() - before[name][k]).double()
            sig = (gr.abs() > 1e-2 * gr.abs().max()).double()
            eu = float(((upd - upd_ref) * sig).norm() / max(float((upd_ref * sig).norm()), 1e-30))
            worst_u = max(worst_u, eu)
            assert eu < 2e-2, (name, k, eu)
    di = iter(d_grads)
    for name in D_NAMES:
        mod = dict(nets[name].named_parameters())
        for k in dn:
            gr = next(di)
            if gr is None or float(gr.abs().max()) < 1e-6:
                continue
            e = float((mod[k].grad.detach().cpu().double() - gr.double()).norm() / gr.double().norm())
            # single-element gradients (the 1-channel output bias) are sums of opposite-signed real/fake terms:
            # cancellation amplifies rounding, so they get a looser bound
            tol = 1e-2 if gr.numel() == 1 else 1e-3
            if gr.numel() > 1:
                worst_g = max(worst_g, e)
            assert e < tol, (name, k, e)
    print("B=%d worst gradient rel-L2 %.3e, worst significant-update rel-L2 %.3e" % (B, worst_g, worst_u))


def _rand_batch(B, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(2):
        real = torch.from_numpy(rs.randn(B, 80, 64).astype(np.float32))
        mask = torch.ones(B, 80, 64)
        for b in range(B):
            size = int(rs.randint(0, 25)); start = int(rs.randint(0, 64 - size))
            mask[b, :, start:start + size] = 0.0
        out += [real, mask]
    return [t.cuda() for t in out]


def test_large_batch_generator_phase_is_the_mean_of_single_sample_phases():
    """BASELINE configs[2] size (bs=32, beyond what the CPU oracle finishes in seconds) through a size-independent
    property: InstanceNorm is per sample and every loss is a batch mean, so the bs=32 generator-phase losses and the
    flat generator gradient must equal the mean over the 32 single-sample phases (which are oracle-checked above).
    Also exercises the generic (non-fused) trunk path: 32 samples x 16 frames > 32 columns."""
    B = 32
    nets = _nets([410 + i for i in range(6)])
    # adversarial terms only: they are smooth, so the linearity is exact up to rounding; the L1 cycle / identity terms are
    # covered against the oracle at bs=1,2 above
    eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=64, cycle_loss_lambda=0.0, identity_loss_lambda=0.0))
    keys = ("g_loss", "adv_loss")

    def g_phase(b):
        if b[0].shape[0] != eng.B:
            eng._use(int(b[0].shape[0]))
        for dst, src in zip(eng.static_in, b):
            dst.copy_(src)
        eng._run_phase("G")
        lo = eng.losses()
        return [lo[k] for k in keys], eng.g_group.grad.double().clone()

    # (the L1 terms are switched off for this property -- see the schedule above -- because |x| has a kink at 0: with 650 k
    # elements per batch some (cycle - real) always sits within rounding of 0, may take the other sign in the differently
    # tiled bs=1 kernels, and ONE flipped sign moves that loss's gradient by 2/sqrt(N) = 0.5 %)
    batch = _rand_batch(B, 77)
    big_l, big_g = g_phase(batch)
    acc_l, acc_g = np.zeros(len(keys)), torch.zeros_like(big_g)
    for i in range(B):
        l1, g1 = g_phase([t[i:i + 1].contiguous() for t in batch])
        acc_l += np.asarray(l1) / B
        acc_g += g1 / B
    for k, a, b in zip(keys, big_l, acc_l):
        assert abs(a - b) < 1e-4 * abs(b), (k, a, b)
    assert np.isfinite(big_l).all()
    # per network (so a small-gradient network is not hidden behind a large one): slice the flat buffers through the
    # parameters' own gradient views
    base = eng.g_group.grad.data_ptr()
    for n in G_NAMES:
        num = den = 0.0
        for p in nets[n].parameters():
            o = (p.grad.data_ptr() - base) // 4
            a, b = big_g[o:o + p.numel()], acc_g[o:o + p.numel()]
            num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
        assert (num / den) ** 0.5 < 1e-3, (n, (num / den) ** 0.5)


def _three_steps(defer, seeds, B=1):
    nets = _nets(seeds)
    eng = TrainEngine(nets, B, 64, schedule=StepSchedule(batch_size=B, n_samples=4 * B))
    eng.defer_d_update = defer
    losses = []
    for it in range(3):
        eng.step(*_rand_batch(B, 900 + it))
        lo = eng.losses()
        losses.append((lo["g_loss"], lo["d_loss"]))
    eng.flush()
    sd = eng.optimizer_state_dict("D")
    return losses, {n: [p.detach().clone() for p in nets[n].parameters()] for n in G_NAMES + D_NAMES}, sd["state"][0]["step"]


def test_deferred_discriminator_update_matches_the_immediate_one(deterministic_mode):
    """Data-parallel ranks defer the discriminator Adam step (and re-pack) to where the discriminators are next used so
    that the gradient all-reduce overlaps the next generator forwards; the arithmetic must not change.  In deterministic
    mode (no floating-point atomics anywhere in the step) that is BIT equality of losses and parameters."""
    seeds = [500 + i for i in range(6)]
    (l0, p0, s0), (l1, p1, s1) = _three_steps(False, seeds), _three_steps(True, seeds)
    assert s0 == s1
    assert l0 == l1, (l0, l1)
    for n in p0:
        for a, b in zip(p0[n], p1[n]):
            assert torch.equal(a, b), n


@pytest.mark.parametrize("B", [1, 8])
def test_step_is_bit_reproducible_in_deterministic_mode(deterministic_mode, B):
    """The reference's CPU path is bit-reproducible run to run.  With MCVC_DETERMINISTIC /
    mcvc_set_deterministic(1) the HIP step is too -- concurrent lanes and auxiliary streams included: three full iterations
    from the same state give identical losses and parameters."""
    seeds = [520 + i for i in range(6)]
    (l0, p0, _), (l1, p1, _) = _three_steps(False, seeds, B), _three_steps(False, seeds, B)
    assert l0 == l1, (l0, l1)
    for n in p0:
        for a, b in zip(p0[n], p1[n]):
            assert torch.equal(a, b), n


@pytest.mark.parametrize("B", [1, 8])
def test_no_result_depends_on_a_float_the_step_did_not_write(deterministic_mode, B):
    """the wide trunk's 1 x 3 implicit GEMMs (more than 64 columns per pass: B = 8) read shifted windows over DENSE rows,
    and the element a window picks up beyond a row's end -- for the first / last row of a tensor a float in FRONT of / BEHIND the tensor,
    i.e. in whatever the allocator handed the neighbouring stash / scratch region -- must contribute exactly 0 whatever it holds.  The
    mask is a bit-and on the loaded value (csrc/sgemm_kernels.hip zmasked; a scale would turn NaN / Inf into NaN).  Here every free block
    of the caching allocator is filled with NaN before the engine allocates its workspaces (torch.empty: the engine's stashes and scratch
    then START as NaN wherever a kernel has not written yet): three iterations must equal the un-poisoned run bit for bit."""
    seeds = [540 + i for i in range(6)]
    l0, p0, _ = _three_steps(False, seeds, B)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((128 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(24)]       # 12 GiB of NaN (bs=8 state + workspaces: 8.3 GB)
    small = [torch.full((n,), float("nan"), device="cuda") for n in (64, 1024, 16384, 262144) for _ in range(64)]   # the small-block pool too
    torch.cuda.synchronize()
    del junk, small
    l1, p1, _ = _three_steps(False, seeds, B)
    assert all(np.isfinite(v) for pair in l1 for v in pair), l1
    assert l0 == l1, (l0, l1)
    for n in p0:
        for a, b in zip(p0[n], p1[n]):
            assert torch.isfinite(b).all(), n
            assert torch.equal(a, b), n


def test_default_mode_stays_close_run_to_run():
    """Default (fast) mode: a few K-split accumulations use atomics, so runs agree to rounding, not bitwise."""
    seeds = [500 + i for i in range(6)]
    (l0, p0, s0), (l1, p1, s1) = _three_steps(False, seeds), _three_steps(True, seeds)
    assert s0 == s1
    assert abs(l0[0][0] - l1[0][0]) < 1e-5 * abs(l0[0][0])
    for a, b in zip(l0, l1):
        assert abs(a[0] - b[0]) < 2e-3 * abs(a[0]) and abs(a[1] - b[1]) < 2e-3 * abs(a[1])
    for n in D_NAMES:
        for a, b in zip(p0[n], p1[n]):
            assert float((a - b).norm()) <= 0.1 * 3e-4 * float(a.numel()) ** 0.5 + 1e-7, n      # 10 % of "every element moved by 3 x lr"


def test_module_forward_after_engine_step_uses_the_updated_weights():
    """the engine's raw-pointer Adam does not bump the parameters' version counters; the modules' packed-weight
    cache must still notice the update (validation / in-process inference call the module API on the same storage)."""
    nets = _nets([540 + i for i in range(6)])
    eng = TrainEngine(nets, 1, 64, schedule=StepSchedule(batch_size=1, n_samples=4))
    x = torch.randn(1, 80, 64, device="cuda"); m = torch.ones_like(x)
    g = nets["generator_A2B"]; d = nets["discriminator_A"]
    with torch.no_grad():
        y_before = g(x, m).clone(); p_before = d(x).clone()       # fills the modules' packed caches
    eng.step(*_rand_batch(1, 33))
    eng.flush()
    with torch.no_grad():
        y_after = g(x, m).clone(); p_after = d(x).clone()
    fresh_g, fresh_d = Generator().cuda(), Discriminator().cuda()
    fresh_g.load_state_dict(g.state_dict()); fresh_d.load_state_dict(d.state_dict())
    with torch.no_grad():
        y_ref = fresh_g(x, m); p_ref = fresh_d(x)
    assert float((y_after - y_before).abs().max()) > 1e-5          # the step did change the weights
    assert float((y_after - y_ref).norm() / y_ref.norm()) < 1e-5, "module forward used stale packed weights"
    assert float((p_after - p_ref).norm() / p_ref.norm()) < 1e-5


def test_trunk_fault_falls_back_to_per_layer_launches_and_training_continues(deterministic_mode):
    """A persistent trunk launch that loses an arrival (test hook mcvc_debug_trunk_fault_inject) poisons its pass with NaN.  The engine's
    fault check must (i) see it, (ii) switch the process to per-layer trunk launches and say so, (iii) leave the engine usable: with the
    parameters and optimizer state of before the poisoned step restored, the following iterations are finite and agree (to the rounding of
    the one iteration that ran on the persistent kernels) with a run that used per-layer launches from the start."""
    from mask_cyclegan_vc._hip import lib
    L = lib()
    B = 1

    def batch(seed):
        rs = np.random.RandomState(seed)
        out = []
        for _ in range(2):
            out.append(torch.from_numpy(rs.randn(B, 80, 64).astype(np.float32)).cuda())
            out.append(torch.from_numpy(orc.fif_mask(rs, B, 80, 64, 25)).cuda())
        return out

    def snapshot(eng):
        eng.flush()
        return [t.clone() for g in (eng.g_group, eng.d_group) for t in (g.flat, g.exp_avg, g.exp_avg_sq)], (eng.g_group.step, eng.d_group.step), \
            (eng.sched.g_opt_lr, eng.sched.d_opt_lr, eng.sched.global_step)

    def restore(eng, snap):
        tensors, steps, sch = snap
        it = iter(tensors)
        for g in (eng.g_group, eng.d_group):
            for t in (g.flat, g.exp_avg, g.exp_avg_sq):
                t.copy_(next(it))
            g.grad.zero_()
        eng.g_group.step, eng.d_group.step = steps
        eng.sched.g_opt_lr, eng.sched.d_opt_lr, eng.sched.global_step = sch
        eng._g_grad_clean = eng._d_grad_clean = True
        eng.repack(G_NAMES + D_NAMES)

    was = L.mcvc_set_trunk_persistent(1)
    try:
        # reference run: per-layer launches from the start
        L.mcvc_set_trunk_persistent(0)
        ref = TrainEngine(_nets([800 + i for i in range(6)]), B, 64, schedule=StepSchedule(batch_size=B, n_samples=4))
        want = []
        for it in range(3):
            ref.step(*batch(40 + it))
            want.append(ref.losses())
        ref.flush()
        # faulting run
        L.mcvc_set_trunk_persistent(1)
        eng = TrainEngine(_nets([800 + i for i in range(6)]), B, 64, schedule=StepSchedule(batch_size=B, n_samples=4))
        assert L.mcvc_gen_trunk_persistent(B, 64) & 1
        eng.step(*batch(40))
        got = [eng.losses()]
        assert eng.check_faults() == 0
        snap = snapshot(eng)
        inj = L.mcvc_debug_trunk_fault_inject(1)
        try:
            eng.step(*batch(41))
            bad = eng.losses()
        finally:
            L.mcvc_debug_trunk_fault_inject(inj)
        assert not np.isfinite(bad["g_loss"]), bad
        with pytest.raises(RuntimeError, match="persistent trunk kernel fault"):
            eng.check_faults()
        assert eng.trunk_fallback and L.mcvc_gen_trunk_persistent(B, 64) == 0          # per-layer launches from here on
        assert eng.check_faults() == 0                                                   # (the error words were cleared)
        restore(eng, snap)
        for it in (1, 2):
            eng.step(*batch(40 + it))
            got.append(eng.losses())
        eng.flush()
        assert all(np.isfinite(v) for lo in got for v in lo.values())
        # iteration 0 ran on the persistent kernels (bit-identical per layer to the per-layer launches up to the statistics' summation order)
        for k in want[0]:
            assert abs(got[0][k] - want[0][k]) <= 1e-5 * abs(want[0][k]) + 1e-8, (k, got[0], want[0])
        for it in (1, 2):
            for k in want[it]:
                assert abs(got[it][k] - want[it][k]) <= 1e-3 * abs(want[it][k]) + 1e-7, (it, k, got[it], want[it])
    finally:
        L.mcvc_set_trunk_persistent(was)
