"""GPU: multi-step parity stated against an fp64 ANCHOR instead of against hand-tuned tolerances.

After several Adam steps a per-tensor comparison of two fp32 implementations measures how often a rounding-level gradient flipped its
sign (Adam's first steps move an element by ~lr * sign(g) whatever |g| is), not whether either is right.  The well-posed statement is the
distance of EACH from the same computation in float64:

    err(X) = || params_X - params_fp64 || / || params_fp64 ||       per parameter tensor, after the four iterations of the `cutoff` fixture
                                                                      (unmodified reference train(), bs = 2, identity cut-off inside)

for X = the reference's own fp32 arithmetic (oracle/ StepOracle in float32: F.conv2d / F.instance_norm on the CPU -- what the fixture was made
with) and X = the HIP step.  The HIP step runs in deterministic mode, so err(HIP) is ONE number per binary and tensor: the gates below are
the measured numbers x 1.25 (and the comparison with the reference's distance, which depends on the CPU's thread count, carries a wider margin).

Both modes of the library are held to "AS ACCURATE AS THE REFERENCE": pooled distance over all parameters within 1.5x of the reference's,
worst tensor and median tensor within 1.5x, no more tensors beyond 1e-3 than the reference has (+2):
  * precise mode (mcvc_set_precise: direct kernels for the 5x5 layers, plain fp32 sums like F.conv2d's);
  * default (fast) mode: the Winograd schemes -- whose batched GEMMs accumulate in two levels since r6 exactly because this test's first
    version measured them 3.6x further from the anchor than the reference.
Reference: mask_cyclegan_vc/train.py:195-315 (the step), :240-242, :297-299 (optimizer steps), :314-315 (identity cut-off)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mcvc_oracle as orc  # noqa: E402
from mask_cyclegan_vc.engine import TrainEngine  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402
from mask_cyclegan_vc.schedule import StepSchedule  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_fp64_anchor import run_cutoff, samples_of  # noqa: E402


def _skip_names(golden_dir):
    norms = json.load(open(os.path.join(golden_dir, "grad_norms.json")))
    return {k.split(":", 1)[1] for k, v in norms.items() if v is not None and v < 1e-6}        # zero-gradient bias class (test_hip_engine.py)


@pytest.fixture(scope="module")
def anchors(golden_dir):
    """(fp64 nets, fp32 reference nets): the oracle on the cutoff fixture in both precisions, once per module (~1 min of host time).  The fp64
    run is first checked against the committed samples (tests/golden/step_cutoff_fp64_samples.npz, made in the build container)."""
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    try:
        o64, l64 = run_cutoff(torch.float64, golden_dir)
        o32, l32 = run_cutoff(torch.float32, golden_dir)
    finally:
        torch.set_num_threads(nt)
    fx = np.load(os.path.join(golden_dir, "step_cutoff_fp64_samples.npz"))
    mine = samples_of(o64)
    for k in mine:        # (fp64 summation order differs between hosts at 1e-16; Adam's first steps amplify that to ~1e-9 on single elements)
        assert np.linalg.norm(mine[k] - fx[k]) <= 1e-6 * max(np.linalg.norm(fx[k]), 1e-30), k
    assert np.allclose(np.asarray(l64), fx["losses"], rtol=1e-10)
    return o64, o32


def _hip_cutoff(golden_dir, precise):
    from mask_cyclegan_vc import _hip
    L = _hip.lib()
    js = json.load(open(os.path.join(golden_dir, "step_cutoff.json")))
    bt = np.load(os.path.join(golden_dir, "step_cutoff_batches.npz"))
    cfg = js["config"]
    was_d, was_p = L.mcvc_set_deterministic(1), L.mcvc_set_precise(1 if precise else 0)
    try:
        nets = {}
        for i, (n, s) in enumerate(zip(orc.NET_ORDER, cfg["filler_seeds"])):
            m = Generator() if i < 2 else Discriminator()
            m.load_state_dict(orc.filler_params("G" if i < 2 else "D", s), strict=True)
            nets[n] = m.cuda()
        sched = StepSchedule(generator_lr=cfg["g_lr"], discriminator_lr=cfg["d_lr"], num_epochs=cfg["num_epochs"], n_samples=cfg["n_utt"],
                             batch_size=cfg["batch_size"], decay_after=cfg["decay_after"], stop_identity_after=cfg["stop_identity_after"])
        eng = TrainEngine(nets, cfg["batch_size"], 64, schedule=sched)
        for it in range(4):
            eng.step(*[torch.from_numpy(bt["it%d_%s" % (it, k)]).cuda() for k in ("real_A", "mask_A", "real_B", "mask_B")])
        eng.flush()
        assert eng.check_faults() == 0
        return {n: {k: p.detach().cpu().clone() for k, p in nets[n].named_parameters()} for n in nets}
    finally:
        L.mcvc_set_deterministic(was_d)
        L.mcvc_set_precise(was_p)


def _distances(golden_dir, got, o64, o32):
    """Per tensor (err_hip, err_ref, numel) against the fp64 anchor + the per-network pooled distances."""
    skip = _skip_names(golden_dir)
    rows, pooled = [], {}
    for name in orc.NET_ORDER:
        pnames = orc.generator_param_names() if name.startswith("gen") else orc.discriminator_param_names()
        nh = nr = dn = 0.0
        for pn in pnames:
            if pn in skip or pn.startswith(orc.DISC_DEAD_PREFIX) or o64[name][pn].numel() == 1:
                continue
            a = o64[name][pn].double()
            dh = float((got[name][pn].double() - a).norm()); dr = float((o32[name][pn].double() - a).norm()); na = float(a.norm())
            rows.append((name, pn, a.numel(), dh / na, dr / na))
            nh += dh * dh; nr += dr * dr; dn += na * na
        pooled[name] = ((nh / dn) ** 0.5, (nr / dn) ** 0.5)
    return rows, pooled


def _report(tag, rows, pooled):
    eh = np.array([r[3] for r in rows]); er = np.array([r[4] for r in rows])
    print("%s vs fp64 anchor: HIP worst %.3e median %.3e beyond 1e-3: %d / %d | reference fp32 worst %.3e median %.3e beyond 1e-3: %d"
          % (tag, eh.max(), np.median(eh), int((eh > 1e-3).sum()), len(eh), er.max(), np.median(er), int((er > 1e-3).sum())))
    for n, (h, r) in pooled.items():
        print("   %-18s pooled distance HIP %.3e  reference %.3e  ratio %.2f" % (n, h, r, h / r))
    worst = sorted(rows, key=lambda r: -r[3])[:4]
    print("   largest: " + "; ".join("%s.%s (%d) %.2e [ref %.2e]" % (r[0], r[1], r[2], r[3], r[4]) for r in worst))
    return eh, er


# The reference's OWN distance to the anchor is a property of the host the oracle runs on (oneDNN / MKL pick their kernels by CPU vendor and
# vector width, the summation order follows the thread count), measured on this fixture:
#     EPYC 9575F, 32 threads (the MI355X boxes):   pooled 5.92e-4, worst tensor 1.08e-3, median 4.89e-5,  4 of 216 tensors beyond 1e-3
#     Xeon (AVX-512), 8 threads (build container): pooled 8.29e-4, worst tensor 1.42e-3, median 6.82e-5, 11 of 216
# (profiles/r06_parity_repeat3.log, profiles/r06_reference_distance_by_host.log).  The HIP side is deterministic -- one number per binary -- so a
# comparison against the LIVE reference alone would pass or fail with the host.  Each reference figure below is therefore the live one
# floored at the MOST ACCURATE reference host measured so far (the first row): on such a host nothing changes, on a host where the reference
# happens to land closer to the anchor the claim stays "as accurate as the reference on the best host we have seen", and on a host where
# it lands further away the live (larger) figure is the honest same-host comparison.
REF_BEST_HOST = {"pooled": {"generator_A2B": 2.300e-4, "generator_B2A": 2.541e-4, "discriminator_A": 2.838e-4, "discriminator_B": 1.344e-4,
                            "discriminator_A2": 1.427e-4, "discriminator_B2": 3.384e-4},
                 "all": 5.924e-4, "worst": 1.078e-3, "median": 4.891e-5, "beyond": 4}


def _as_accurate_as_the_reference(eh, er, pooled):
    """The statement "as accurate as the reference's fp32 step", on one fixture.  The quantity is chaotic (a flipped rounding-level sign in
    iteration 1 changes which signs flip in iteration 2), so single networks scatter by ~2x either way between arithmetically equivalent
    implementations -- the reference at another thread count or on another CPU included; the gates are on the whole step, with that scatter
    allowed per network, and the reference's figures are floored as REF_BEST_HOST says."""
    num = sum(h * h for h, _ in pooled.values()) ** 0.5
    den_live = sum(r * r for _, r in pooled.values()) ** 0.5
    den = max(den_live, REF_BEST_HOST["all"])
    print("   all networks: HIP %.3e reference %.3e ratio %.2f" % (num, den_live, num / den_live))
    assert num <= 1.5 * den, (num, den_live)
    for n, (h, r) in pooled.items():
        assert h <= 3.0 * max(r, REF_BEST_HOST["pooled"][n]), (n, h, r)
    assert eh.max() <= 1.5 * max(er.max(), REF_BEST_HOST["worst"]), (eh.max(), er.max())
    assert np.median(eh) <= 1.5 * max(np.median(er), REF_BEST_HOST["median"]), (np.median(eh), np.median(er))
    assert int((eh > 1e-3).sum()) <= max(int((er > 1e-3).sum()), REF_BEST_HOST["beyond"]) + 2


def test_precise_mode_is_as_accurate_as_the_reference_fp32(golden_dir, anchors):
    o64, o32 = anchors
    rows, pooled = _distances(golden_dir, _hip_cutoff(golden_dir, precise=True), o64, o32)
    eh, er = _report("precise mode", rows, pooled)
    _as_accurate_as_the_reference(eh, er, pooled)
    # deterministic mode: exact repeats.  Measured with the r6 binary (profiles/r06_parity_fp64.log); gates = measured x 1.25
    assert eh.max() <= 1.141e-3 * 1.25 and np.median(eh) <= 4.984e-5 * 1.25 and int((eh > 1e-3).sum()) <= 3 + 1


def test_default_mode_is_as_accurate_as_the_reference_fp32(golden_dir, anchors):
    """The default (fast) mode: Winograd F(2x2 / 4x4, 5x5 / 3x3) on the 5x5 layers.  Until r6 its K-long fp32 accumulation chains in the
    Winograd domain put it 3.6x (median over tensors) further from the anchor than the reference is (worst tensor 3.8e-3, 65 of 216 beyond
    1e-3: profiles/r06_parity_probe_before.log); with the two-level accumulation of the batched GEMMs (csrc/wino_kernels.hip kFoldK) the
    schemes' op-level error fell 2.2-3.6x and the step sits where the reference sits."""
    o64, o32 = anchors
    rows, pooled = _distances(golden_dir, _hip_cutoff(golden_dir, precise=False), o64, o32)
    eh, er = _report("default (fast) mode", rows, pooled)
    _as_accurate_as_the_reference(eh, er, pooled)
    assert eh.max() <= 7.201e-4 * 1.25 and np.median(eh) <= 3.122e-5 * 1.25 and int((eh > 1e-3).sum()) <= 0 + 1
