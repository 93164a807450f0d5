"""fp64 anchor of the `cutoff` fixture (tests/golden/step_cutoff*): oracle.StepOracle run in float64 on the fixture's four recorded
minibatches with the reference's bookkeeping (identity lambda 5, 5, 0, 0: train.py:314-315) -> per parameter tensor the values at
oracle.sample_index positions (the positions the fixture itself samples) and the tensor's l2 norm, in float64.

  python tests/golden/make_fp64_anchor.py        (CPU, ~2.5 min on 8 cores; needs only oracle/ and the committed fixture)

The anchor is to fp32 rounding what an exact answer is: both the reference's fp32 arithmetic and the HIP step are measured against it
(tests/test_oracle_golden.py::test_reference_fp32_against_the_fp64_anchor, tests/test_hip_parity_fp64.py).  The GPU tests recompute the
fp64 run in full on the box (they need whole tensors) and check it against these samples first."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import mcvc_oracle as orc  # noqa: E402


def run_cutoff(dtype, golden_dir=HERE, n_it=4):
    """(nets, losses): the oracle on the cutoff fixture's batches in `dtype`."""
    js = json.load(open(os.path.join(golden_dir, "step_cutoff.json")))
    bt = np.load(os.path.join(golden_dir, "step_cutoff_batches.npz"))
    cfg = js["config"]
    nets = {n: orc.filler_params("G" if i < 2 else "D", s, dtype=dtype) for i, (n, s) in enumerate(zip(orc.NET_ORDER, cfg["filler_seeds"]))}
    so = orc.StepOracle(nets, g_lr=cfg["g_lr"], d_lr=cfg["d_lr"], skip_wasted=True)
    lam = [t["identity_lambda_before_check"] for t in js["trace"]]
    losses = []
    for it in range(n_it):
        batch = [torch.from_numpy(bt["it%d_%s" % (it, k)]).to(dtype) for k in ("real_A", "mask_A", "real_B", "mask_B")]
        so.identity_lambda = float(lam[it])
        losses.append(so.step(*batch))
    return nets, losses


def samples_of(nets):
    out = {}
    for name in orc.NET_ORDER:
        pnames = orc.generator_param_names() if name.startswith("gen") else orc.discriminator_param_names()
        for j, pn in enumerate(pnames):
            if pn.startswith(orc.DISC_DEAD_PREFIX):
                continue
            t = nets[name][pn].detach().double().flatten()
            out["s_%s_%d" % (name, j)] = t[torch.from_numpy(orc.sample_index(t.numel()))].numpy()
            out["n_%s_%d" % (name, j)] = np.float64(t.norm())
    return out


if __name__ == "__main__":
    nets, losses = run_cutoff(torch.float64)
    out = samples_of(nets)
    out["losses"] = np.asarray(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "step_cutoff_fp64_samples.npz"), **out)
    print("losses", losses)
