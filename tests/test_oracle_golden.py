"""Pin the CPU oracle (oracle/mcvc_oracle.py) to vectors produced by the reference itself.

Fixtures: tests/golden/*.npz|json, written by tests/golden/make_golden.py from /root/reference.
The oracle and the reference both call ATen on CPU, so agreement is expected to ~1e-6; the gates
below are 1e-5 relative L2 (forward / grads) and 2e-4 on multi-step parameter norms.
"""
import json
import os

import numpy as np
import pytest
import torch

import mcvc_oracle as orc


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def seeded_inputs(seed, B, T, max_mask_len=25):
    rs = np.random.RandomState(seed)
    x = rs.randn(B, 80, T).astype(np.float32)
    m = orc.fif_mask(rs, B, 80, T, min(max_mask_len, T))
    return torch.from_numpy(x), torch.from_numpy(m)


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.load(open(os.path.join(golden_dir, "meta.json")))


def test_key_layout_matches_reference(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "layout.json")))
    g = orc.generator_key_shapes()
    assert [[k, list(v)] for k, v in g.items()] == lay["generator_state_dict"]
    assert len(g) == 114
    assert orc.generator_param_names() == lay["generator_named_parameters"]
    d = orc.discriminator_key_shapes()
    assert [[k, list(v)] for k, v in d.items()] == lay["discriminator_state_dict"]
    assert len(d) == 20
    assert orc.discriminator_param_names() == lay["discriminator_named_parameters"]
    assert lay["instance_norm_eps"] == orc.IN_EPS
    assert lay["instance_norm_track_running_stats"] is False


def test_default_init_matches_reference_seed0(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "layout.json")))["default_init_seed0"]
    nets = orc.default_init_nets(0)
    for name in orc.NET_ORDER:
        names = orc.generator_param_names() if name.startswith("generator") else orc.discriminator_param_names()
        ps = [nets[name][k] for k in names]
        s = float(sum(p.double().abs().sum() for p in ps))
        assert abs(s - lay[name]["abs_sum"]) <= 1e-9 * lay[name]["abs_sum"], name
        assert [float(v) for v in ps[0].flatten()[:4]] == lay[name]["first"]
        assert [float(v) for v in ps[-2].flatten()[:4]] == lay[name]["last"]


def test_forward_matches_reference(golden_dir, meta):
    gold = np.load(os.path.join(golden_dir, "forward.npz"))
    g = orc.filler_params("G", meta["filler_seeds"]["G"])
    d = orc.filler_params("D", meta["filler_seeds"]["D"])
    with torch.no_grad():
        for case in meta["forward_cases"]:
            B, T = case["B"], case["T"]
            x, m = seeded_inputs(case["seed"], B, T)
            y = orc.generator_forward(g, x, m)
            tag = "%dx%d" % (B, T)
            assert y.shape == gold["g_out_" + tag].shape
            assert rel_l2(y.numpy(), gold["g_out_" + tag]) < 1e-5
            assert rel_l2(orc.discriminator_forward(d, x).numpy(), gold["d_out_" + tag]) < 1e-5
            assert rel_l2(orc.discriminator_forward(d, y).numpy(), gold["dg_out_" + tag]) < 1e-5


def test_layer_taps_match_reference_digests(golden_dir, meta):
    dig = json.load(open(os.path.join(golden_dir, "layer_digests.json")))
    g = orc.filler_params("G", meta["filler_seeds"]["G"])
    d = orc.filler_params("D", meta["filler_seeds"]["D"])
    x, m = seeded_inputs(1000, 1, 64)
    gt, dt = {}, {}
    with torch.no_grad():
        y = orc.generator_forward(g, x, m, taps=gt)
        orc.discriminator_forward(d, y, taps=dt)
    # reference leaf outputs that coincide with oracle tap points
    pairs = [
        ("G:downSample2.convLayer_gates.1", None),          # presence check only
        ("G:conv2dto1dLayer_tfan", gt["conv2dto1d"]),
        ("G:residualLayer6.conv1d_out_layer.1", None),
        ("G:conv1dto2dLayer_tfan", gt["conv1dto2d"].reshape(1, 5120, -1)),
        ("D:downSample3.2", dt["downSample3"]),
        ("D:convLayer1.1", dt["convLayer1"]),
    ]
    for key, t in pairs:
        assert key in dig, key
        if t is None:
            continue
        ref = dig[key][0]
        assert list(t.shape) == ref["shape"]
        assert abs(float(t.double().mean()) - ref["mean"]) < 1e-5 * max(1.0, abs(ref["mean"]))
        assert abs(float(t.double().std()) - ref["std"]) < 1e-5 * ref["std"]
        np.testing.assert_allclose(t.flatten()[:8].numpy(), np.array(ref["first"], dtype=np.float32), rtol=2e-4, atol=2e-5)
    # the aliased upSample2 module fires once per call, under both names in the digest
    assert "G:upSample2.3" in dig or "G:convLayer.3" in dig


def test_gradients_match_reference(golden_dir, meta):
    gold = np.load(os.path.join(golden_dir, "grads.npz"))
    norms = json.load(open(os.path.join(golden_dir, "grad_norms.json")))
    g = orc.filler_params("G", meta["filler_seeds"]["G"])
    d = orc.filler_params("D", meta["filler_seeds"]["D"])
    c = meta["grad_case"]
    x, m = seeded_inputs(c["seed"], c["B"], c["T"])
    x.requires_grad_(True)
    gn, dn = orc.generator_param_names(), orc.discriminator_param_names()
    gl = [g[k].requires_grad_(True) for k in gn]
    dl = [d[k].requires_grad_(True) for k in dn]
    y = orc.generator_forward(g, x, m)
    loss = torch.mean((1 - orc.discriminator_forward(d, y)) ** 2) + 10.0 * torch.mean(torch.abs(x.detach() - y))
    grads = torch.autograd.grad(loss, [x] + gl + dl, allow_unused=True)
    assert abs(float(loss) - float(gold["loss"])) < 1e-5 * abs(float(gold["loss"]))
    assert rel_l2(grads[0].numpy(), gold["dx"]) < 1e-4
    checked = 0
    for tag, names, gs in (("G", gn, grads[1:1 + len(gn)]), ("D", dn, grads[1 + len(gn):])):
        for n, gr in zip(names, gs):
            ref = norms[tag + ":" + n]
            if ref is None:
                assert gr is None and n.startswith(orc.DISC_DEAD_PREFIX)
                continue
            if ref < 1e-6:           # conv biases in front of an InstanceNorm: mathematically zero
                assert float(gr.norm()) < 1e-5
                continue
            assert abs(float(gr.double().norm()) - ref) < 2e-4 * ref, n
            assert rel_l2(gr.flatten()[:32].numpy(), gold[tag + ":" + n]) < 5e-3, n
            checked += 1
    assert checked > 80


def _zero_bias(golden_dir):
    """Conv biases in front of an InstanceNorm: their gradient is mathematically zero (the reference's own grad_norms.json says < 1e-6)."""
    return {k.split(":", 1)[1] for k, v in json.load(open(os.path.join(golden_dir, "grad_norms.json"))).items() if v is not None and v < 1e-6}


def _norm_gate(tol, rn, k, numel, it, lr, zero_bias):
    """Allowed |norm - fixture norm| of one parameter tensor after ``it + 1`` optimizer steps.

    Well-posed tensors: ``tol`` relative.  Two classes are NOT well-posed, on ANY host: the conv biases in front of an InstanceNorm (true
    gradient 0: what reaches Adam is rounding residue) and the one-element biases (a cancelling sum).  Adam's first steps move an element by
    ~lr * sign(g) whatever |g| is, so a residue whose sign depends on the CPU's summation order (vector width, thread count, oneDNN kernel
    choice) becomes a +-lr difference per element and step.  The fixtures were written on one host and this test runs on whatever host the
    suite is given (seen: the same oracle, same thread count, AVX-512 Xeon against the fixture's host -- 3.8e-4 absolute on a 256-element
    bias after ONE step, every well-posed tensor at 1e-7).  Their gate is therefore Adam's own bound on what those flips can do to the
    norm: ||dp|| <= steps * lr * sqrt(numel)."""
    if k in zero_bias or numel == 1:
        return (it + 1) * lr * numel ** 0.5
    return tol * max(rn, 1e-3)


def _load_step(golden_dir, tag):
    js = json.load(open(os.path.join(golden_dir, "step_%s.json" % tag)))
    bt = np.load(os.path.join(golden_dir, "step_%s_batches.npz" % tag))
    return js, bt


def _nets_from_filler(seeds):
    return {n: orc.filler_params("G" if i < 2 else "D", s) for i, (n, s) in enumerate(zip(orc.NET_ORDER, seeds))}


@pytest.mark.parametrize("skip_wasted", [False, True])
def test_full_step_matches_unmodified_reference_train(golden_dir, skip_wasted):
    """3 iterations of the reference's unmodified train() (bs=1) vs StepOracle on the recorded batches."""
    js, bt = _load_step(golden_dir, "plain")
    nets = _nets_from_filler(js["config"]["filler_seeds"])
    so = orc.StepOracle(nets, skip_wasted=skip_wasted)
    n_it = 3 if not skip_wasted else 2
    zb, lr = _zero_bias(golden_dir), js["config"]["g_lr"]
    for it in range(n_it):
        batch = [torch.from_numpy(bt["it%d_%s" % (it, k)]) for k in ("real_A", "mask_A", "real_B", "mask_B")]
        g_loss, d_loss = so.step(*batch)
        assert abs(g_loss - js["losses"][it]["g_loss"]) < 2e-4 * abs(js["losses"][it]["g_loss"])
        assert abs(d_loss - js["losses"][it]["d_loss"]) < 2e-4 * abs(js["losses"][it]["d_loss"])
        ref_norms = js["trace"][it]["norms"]
        for name in orc.NET_ORDER:
            names = orc.generator_param_names() if name.startswith("generator") else orc.discriminator_param_names()
            for k, rn in zip(names, ref_norms[name]):
                mine = float(nets[name][k].double().norm())
                assert abs(mine - rn) <= _norm_gate(2e-4, rn, k, nets[name][k].numel(), it, lr, zb), (it, name, k)
    # Adam state exists only for parameters that ever received a grad (D downSample4 never does)
    dn = orc.discriminator_param_names()
    dead = {n * len(dn) + i for n in range(4) for i, k in enumerate(dn) if k.startswith(orc.DISC_DEAD_PREFIX)}
    assert sorted(set(range(4 * len(dn))) - dead) == js["adam_state_keys_D"]
    assert sorted(so.d_opt.state.keys()) == js["adam_state_keys_D"]
    assert sorted(so.g_opt.state.keys()) == js["adam_state_keys_G"] == list(range(220))


def _replay(golden_dir, tag, n_it, tol):
    """StepOracle over a fixture of the unmodified train(), with the reference's end-of-iteration bookkeeping (train.py:307-315) applied
    to the oracle's learning rates and identity weight between iterations."""
    js, bt = _load_step(golden_dir, tag)
    cfg = js["config"]
    nets = _nets_from_filler(cfg["filler_seeds"])
    so = orc.StepOracle(nets, skip_wasted=True)
    g_lr, d_lr, gs = cfg["g_lr"], cfg["d_lr"], 0
    denom = float(cfg["num_epochs"] * (cfg["n_utt"] // cfg["batch_size"]))
    zb = _zero_bias(golden_dir)
    for it in range(n_it):
        batch = [torch.from_numpy(bt["it%d_%s" % (it, k)]) for k in ("real_A", "mask_A", "real_B", "mask_B")]
        assert js["trace"][it]["identity_lambda_before_check"] == so.identity_lambda
        g_loss, d_loss = so.step(*batch)
        assert abs(g_loss - js["losses"][it]["g_loss"]) < tol * abs(js["losses"][it]["g_loss"]), (it, g_loss)
        assert abs(d_loss - js["losses"][it]["d_loss"]) < tol * abs(js["losses"][it]["d_loss"]), (it, d_loss)
        for name in orc.NET_ORDER:
            names = orc.generator_param_names() if name.startswith("generator") else orc.discriminator_param_names()
            for k, rn in zip(names, js["trace"][it]["norms"][name]):
                assert abs(float(nets[name][k].double().norm()) - rn) <= _norm_gate(tol, rn, k, nets[name][k].numel(), it, cfg["g_lr"], zb), (it, name, k)
        gs += cfg["batch_size"]
        if gs > cfg["decay_after"]:                                   # train.py:307-311, call-site bug included
            g_lr = max(0.0, g_lr - cfg["g_lr"] / denom)
            d_lr = max(0.0, d_lr - cfg["d_lr"] / denom)
            so.g_opt.lr = d_lr
        if gs > cfg["stop_identity_after"]:                           # train.py:314-315
            so.identity_lambda = 0
    return js, bt, nets


def test_oracle_across_the_lr_decay_bug(golden_dir):
    _replay(golden_dir, "decay", 2, 2e-4)


def test_oracle_past_the_identity_cutoff(golden_dir):
    """Iterations 2 and 3 of the ``cutoff`` fixture run with identity_loss_lambda == 0 in the reference's unmodified train()
    (train.py:207-210 still computes the identity forwards, :223-224 weighs them 0).  Also pins the fixture's parameter SAMPLES:
    element-wise values after four iterations, not only norms.  The element gate is the reference's own arithmetic spread after four Adam
    steps (test_reference_arithmetic_spread_after_four_adam_steps asserts the same 3.5e-3 between thread counts): on the fixture's host this
    replay sits under 1e-3; on a host whose vector width differs from it (AVX-512 Xeon) the SAME code is 2.0e-3 away on upSample1 / downSample2
    with every loss within 4e-5 and every well-posed norm within 4e-5."""
    js, bt, nets = _replay(golden_dir, "cutoff", 4, 3e-4)
    assert [t["identity_lambda_before_check"] for t in js["trace"]] == [5, 5, 0, 0] and js["final"]["identity_loss_lambda"] == 0
    zero_bias = _zero_bias(golden_dir)
    for name in orc.NET_ORDER:
        names = orc.generator_param_names() if name.startswith("generator") else orc.discriminator_param_names()
        for j, k in enumerate(names):
            if k in zero_bias:
                continue
            flat = nets[name][k].flatten()
            mine = flat[torch.from_numpy(orc.sample_index(flat.numel()))].numpy()
            ref = bt["final_%s_%d" % (name, j)]
            if ref.size == 1:            # the discriminators' one-element output bias: cancellation + Adam's normalisation (see test_hip_engine)
                continue
            assert rel_l2(mine, ref) < 7e-3 / 2, (name, k, rel_l2(mine, ref))


def test_reference_arithmetic_spread_after_four_adam_steps(golden_dir):
    """How far the REFERENCE's own arithmetic is from itself after the four iterations of the ``cutoff`` fixture when only the summation
    order changes (3 threads and the full autograd graph against the fixture's 8 threads): Adam's first steps move an element by
    ~lr * sign(g) whatever |g| is, so rounding-level gradients turn into +-lr parameter differences.  This spread -- not 1e-3 -- is what
    bounds per-tensor parity of ANY implementation after several optimizer steps; tests/test_hip_engine.py gates the HIP engine at 2 x
    the upper bound asserted here, and at 1e-3 on everything that is well-posed (losses, norms, single-iteration gradients)."""
    js, bt = _load_step(golden_dir, "cutoff")
    was = torch.get_num_threads()
    torch.set_num_threads(3)
    try:
        nets = _nets_from_filler(js["config"]["filler_seeds"])
        so = orc.StepOracle(nets, skip_wasted=False)
        for it, lam in enumerate((5, 5, 0, 0)):
            so.identity_lambda = float(lam)
            g_loss, d_loss = so.step(*[torch.from_numpy(bt["it%d_%s" % (it, k)]) for k in ("real_A", "mask_A", "real_B", "mask_B")])
            assert abs(g_loss - js["losses"][it]["g_loss"]) < 1e-3 * abs(js["losses"][it]["g_loss"])     # (observed 4e-4 at the fourth iteration)
    finally:
        torch.set_num_threads(was)
    zero_bias = _zero_bias(golden_dir)
    errs = []
    for name in orc.NET_ORDER:
        names = orc.generator_param_names() if name.startswith("generator") else orc.discriminator_param_names()
        for j, k in enumerate(names):
            ref = bt["final_%s_%d" % (name, j)]
            if k in zero_bias or ref.size == 1:
                continue
            flat = nets[name][k].flatten()
            errs.append(rel_l2(flat[torch.from_numpy(orc.sample_index(flat.numel()))].numpy(), ref))
    errs = np.asarray(errs)
    print("reference self-spread after 4 steps: worst %.2e, %d of %d sampled tensors beyond 1e-3" % (errs.max(), int((errs > 1e-3).sum()), errs.size))
    assert errs.max() < 7e-3 / 2                   # the bound the GPU test scales
    assert errs.max() > 1e-3                       # ... and the reason a 1e-3 per-tensor gate after several steps is not a property of the path


def test_reference_fp32_against_the_fp64_anchor(golden_dir):
    """The anchor that replaces hand-tuned multi-step tolerances: ``tests/golden/step_cutoff_fp64_samples.npz`` is the
    cutoff fixture's four iterations computed in FLOAT64 (tests/golden/make_fp64_anchor.py; same batches, same bookkeeping).  Here the
    REFERENCE's own fp32 result -- the fixture's element samples, written by the unmodified train() -- is measured against it: the distance
    every fp32 implementation of this step has to be judged by (tests/test_hip_parity_fp64.py judges the HIP step the same way, on whole
    tensors).  No oracle run: fixture against fixture."""
    js, bt = _load_step(golden_dir, "cutoff")
    fx = np.load(os.path.join(golden_dir, "step_cutoff_fp64_samples.npz"))
    for it in range(4):       # fp32 losses vs fp64 losses: rounding only before the first update, the steps' amplification of it afterwards
        assert abs(fx["losses"][it][0] - js["losses"][it]["g_loss"]) < (2e-6 if it == 0 else 2e-4) * abs(js["losses"][it]["g_loss"])
        assert abs(fx["losses"][it][1] - js["losses"][it]["d_loss"]) < (2e-6 if it == 0 else 1e-3) * abs(js["losses"][it]["d_loss"])
    zero_bias = _zero_bias(golden_dir)
    errs, pooled = [], {}
    for name in orc.NET_ORDER:
        names = orc.generator_param_names() if name.startswith("generator") else orc.discriminator_param_names()
        num = den = 0.0
        for j, k in enumerate(names):
            key = "s_%s_%d" % (name, j)
            if k in zero_bias or k.startswith(orc.DISC_DEAD_PREFIX) or fx[key].size == 1:
                continue
            ref32 = bt["final_%s_%d" % (name, j)].astype(np.float64)
            d = float(np.linalg.norm(ref32 - fx[key])); n = float(np.linalg.norm(fx[key]))
            errs.append(d / n)
            num += d * d; den += n * n
            # the fixture's per-tensor norms (whole tensors) against the anchor's
            assert abs(js["trace"][-1]["norms"][name][j] - float(fx["n_%s_%d" % (name, j)])) <= 1e-3 * float(fx["n_%s_%d" % (name, j)])
        pooled[name] = (num / den) ** 0.5
    errs = np.asarray(errs)
    print("reference fp32 (fixture samples) vs fp64 anchor: worst %.2e median %.2e, %d of %d beyond 1e-3; pooled per network %s"
          % (errs.max(), np.median(errs), int((errs > 1e-3).sum()), errs.size, {k: "%.2e" % v for k, v in pooled.items()}))
    assert errs.max() < 4e-3 and np.median(errs) < 2e-4
    assert max(pooled.values()) < 5e-5


def test_dataset_mask_law(golden_dir):
    """FIF masks drawn by the reference VCDataset: ones with one zeroed span shared by all 80 bins."""
    dr = np.load(os.path.join(golden_dir, "dataset_draws.npz"))
    for k in range(4):
        for nm in ("mA", "mB"):
            m = dr["d%d_%s" % (k, nm)]
            assert m.shape == (80, 64)
            assert set(np.unique(m)).issubset({0.0, 1.0})
            col = m[0]
            assert (m == col[None, :]).all()
            zeros = np.where(col == 0)[0]
            if len(zeros):
                assert len(zeros) < 25 and zeros[-1] - zeros[0] + 1 == len(zeros)
