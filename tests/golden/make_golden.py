#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container, where the reference is mounted read-only at /root/reference; the
reference's source never travels -- only the small .npz / .json vectors written here are committed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Recipe: stub the non-hot-path imports (cv2, torchaudio, torchvision, librosa,
tensorboardX), import ``mask_cyclegan_vc.{model,train}`` from /root/reference unmodified, drive
``Generator`` / ``Discriminator`` / ``MaskCycleGANVCTraining.train()`` on seeded inputs and record
outputs.  Weights come from ``oracle.mcvc_oracle.filler_params`` (numpy RandomState, so fixtures do
not depend on the torch RNG implementation) and are loaded with ``load_state_dict(strict=True)``,
which also pins the 114 / 20 key layout.
"""
import json
import os
import random
import sys
import tempfile
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mcvc_oracle as orc  # noqa: E402


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _SW:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    mod("cv2")
    mod("torchaudio")
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", ToTensor=object)
    tv.utils = mod("torchvision.utils")
    lb = mod("librosa")
    lb.display = mod("librosa.display")
    mod("tensorboardX", SummaryWriter=_SW)


def seeded_inputs(seed, B, T, max_mask_len=25):
    rs = np.random.RandomState(seed)
    x = rs.randn(B, 80, T).astype(np.float32)
    m = orc.fif_mask(rs, B, 80, T, min(max_mask_len, T))
    return torch.from_numpy(x), torch.from_numpy(m)


# ---- 6. unmodified train() ---------------------------------------------------------------
def run_train(refs, decay_after, stop_identity_after, n_utt, bs, tag, sample_final=False):
    Generator, Discriminator, MaskCycleGANVCTraining, VCDataset, TrainLogger = refs
    seed = 0
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "gold"), exist_ok=True)
    rs = np.random.RandomState(1234)
    data_A = [rs.randn(80, 64 + rs.randint(0, 40)).astype(np.float32) for _ in range(n_utt)]
    data_B = [rs.randn(80, 64 + rs.randint(0, 40)).astype(np.float32) for _ in range(n_utt)]
    t = object.__new__(MaskCycleGANVCTraining)
    t.num_epochs = 1
    t.start_epoch = 1
    t.generator_lr = 2e-4
    t.discriminator_lr = 1e-4
    t.decay_after = decay_after
    t.stop_identity_after = stop_identity_after
    t.mini_batch_size = bs
    t.cycle_loss_lambda = 10
    t.identity_loss_lambda = 5
    t.device = "cpu"
    t.epochs_per_save = 10 ** 9
    t.epochs_per_plot = 10 ** 9
    t.n_samples = n_utt
    t.generator_lr_decay = t.generator_lr / float(t.num_epochs * (t.n_samples // t.mini_batch_size))
    t.discriminator_lr_decay = t.discriminator_lr / float(t.num_epochs * (t.n_samples // t.mini_batch_size))
    t.dataset = VCDataset(datasetA=data_A, datasetB=data_B, n_frames=64, max_mask_len=25)

    class RecordingLoader:
        """Wraps the DataLoader the harness hands to the unmodified train(); records batches."""

        def __init__(self, inner):
            self.inner = inner
            self.dataset = inner.dataset
            self.batches = []

        def __iter__(self):
            for b in self.inner:
                self.batches.append([np.asarray(v).copy() for v in b])
                yield b

        def __len__(self):
            return len(self.inner)

    t.train_dataloader = RecordingLoader(torch.utils.data.DataLoader(
        dataset=t.dataset, batch_size=bs, shuffle=True, drop_last=False))
    largs = Namespace(batch_size=bs, save_dir=tmp, name="gold", start_epoch=1, steps_per_print=1, num_epochs=1)
    t.logger = TrainLogger(largs, len(t.dataset))
    t.saver = None
    names = list(orc.NET_ORDER)
    nets = [Generator(), Generator(), Discriminator(), Discriminator(), Discriminator(), Discriminator()]
    for i, (n, net) in enumerate(zip(names, nets)):
        net.load_state_dict(orc.filler_params("G" if i < 2 else "D", 300 + i), strict=True)
        setattr(t, n, net)
    g_params = list(t.generator_A2B.parameters()) + list(t.generator_B2A.parameters())
    d_params = (list(t.discriminator_A.parameters()) + list(t.discriminator_B.parameters())
                + list(t.discriminator_A2.parameters()) + list(t.discriminator_B2.parameters()))
    t.generator_optimizer = torch.optim.Adam(g_params, lr=t.generator_lr, betas=(0.5, 0.999))
    t.discriminator_optimizer = torch.optim.Adam(d_params, lr=t.discriminator_lr, betas=(0.5, 0.999))

    # observe (not modify) per-iteration state through the logger's end_iter, which train() calls
    trace = []
    orig_end_iter = t.logger.end_iter

    def end_iter():
        orig_end_iter()
        trace.append({
            "global_step": t.logger.global_step,
            "g_opt_lr": t.generator_optimizer.param_groups[0]["lr"],
            "d_opt_lr": t.discriminator_optimizer.param_groups[0]["lr"],
            "identity_lambda_before_check": t.identity_loss_lambda,
            "norms": {n: [float(p.detach().double().norm()) for p in getattr(t, n).parameters()] for n in names},
        })
    t.logger.end_iter = end_iter
    losses = []
    orig_log_iter = t.logger.log_iter

    def log_iter(loss_dict):
        losses.append(dict(loss_dict))
        orig_log_iter(loss_dict)
    t.logger.log_iter = log_iter

    t.train()
    out = {"losses": losses,
           "trace": trace,
           "final": {"generator_lr_attr": t.generator_lr, "discriminator_lr_attr": t.discriminator_lr,
                     "identity_loss_lambda": t.identity_loss_lambda,
                     "g_opt_lr": t.generator_optimizer.param_groups[0]["lr"],
                     "d_opt_lr": t.discriminator_optimizer.param_groups[0]["lr"]},
           "adam_state_keys_G": sorted(t.generator_optimizer.state_dict()["state"].keys()),
           "adam_state_keys_D": sorted(t.discriminator_optimizer.state_dict()["state"].keys()),
           "adam_group_keys": sorted(t.generator_optimizer.state_dict()["param_groups"][0].keys()),
           "config": {"decay_after": decay_after, "stop_identity_after": stop_identity_after,
                      "n_utt": n_utt, "batch_size": bs, "filler_seeds": [300 + i for i in range(6)],
                      "g_lr": 2e-4, "d_lr": 1e-4, "num_epochs": 1}}
    batches = {}
    if sample_final:
        # a fixed sample of every parameter tensor after the last iteration (full tensors would be 590 MB): pins the
        # multi-step UPDATE element by element, not only its norm
        for n in names:
            for j, p in enumerate(getattr(t, n).parameters()):
                flat = p.detach().flatten()
                batches["final_%s_%d" % (n, j)] = flat[torch.from_numpy(orc.sample_index(flat.numel()))].numpy().copy()
    for i, b in enumerate(t.train_dataloader.batches):
        for nm, arr in zip(("real_A", "mask_A", "real_B", "mask_B"), b):
            batches["it%d_%s" % (i, nm)] = arr.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "step_%s_batches.npz" % tag), **batches)
    json.dump(out, open(os.path.join(HERE, "step_%s.json" % tag), "w"), indent=0)
    print(tag, losses)


TRAIN_CASES = {
    "plain": dict(decay_after=1e9, stop_identity_after=1e9, n_utt=3, bs=1),
    # trip the LR-decay call-site bug (train.py:307-311) and the identity cut-off (train.py:314-315) early
    "decay": dict(decay_after=1, stop_identity_after=2, n_utt=4, bs=2),
    # iterations 2 and 3 RUN with identity_loss_lambda == 0 (train.py:314-315 trips after iteration 1; :207-210 keeps computing the
    # identity forwards): the regime a canonical run spends > 97 % of its life in
    "cutoff": dict(decay_after=1e9, stop_identity_after=2, n_utt=8, bs=2, sample_final=True),
}


def run_train_case(tag, *refs):
    run_train(refs, tag=tag, **TRAIN_CASES[tag])


def main(only=None):
    torch.set_num_threads(8)
    _stub_modules()
    sys.path.insert(0, REF)
    from mask_cyclegan_vc.model import Generator, Discriminator  # reference, unmodified
    from mask_cyclegan_vc.train import MaskCycleGANVCTraining    # reference, unmodified
    from dataset.vc_dataset import VCDataset                     # reference, unmodified
    from logger.train_logger import TrainLogger                  # reference, unmodified

    meta = {"torch": torch.__version__, "numpy": np.__version__}

    if only is not None:           # regenerate ONE train() fixture without touching the others
        return run_train_case(only, Generator, Discriminator, MaskCycleGANVCTraining, VCDataset, TrainLogger)

    # ---- 1. key layout --------------------------------------------------------------------
    g, d = Generator(), Discriminator()
    layout = {
        "generator_state_dict": [[k, list(v.shape)] for k, v in g.state_dict().items()],
        "generator_named_parameters": [n for n, _ in g.named_parameters()],
        "discriminator_state_dict": [[k, list(v.shape)] for k, v in d.state_dict().items()],
        "discriminator_named_parameters": [n for n, _ in d.named_parameters()],
        "generator_children": [n for n, _ in g.named_children()],
        "instance_norm_eps": g.conv2dto1dLayer_tfan.eps,
        "instance_norm_track_running_stats": g.conv2dto1dLayer_tfan.track_running_stats,
    }

    # ---- 2. default-init pin (torch.manual_seed(0), construction order of train.py:103-110) --
    torch.manual_seed(0)
    nets0 = [Generator(), Generator(), Discriminator(), Discriminator(), Discriminator(), Discriminator()]
    layout["default_init_seed0"] = {
        name: {"abs_sum": float(sum(p.double().abs().sum() for p in n.parameters())),
               "first": [float(v) for v in next(iter(n.parameters())).flatten()[:4]],
               "last": [float(v) for v in list(n.parameters())[-2].flatten()[:4]]}
        for name, n in zip(orc.NET_ORDER, nets0)}

    # ---- 3. forward goldens -----------------------------------------------------------------
    g.load_state_dict(orc.filler_params("G", 101), strict=True)
    d.load_state_dict(orc.filler_params("D", 202), strict=True)
    fwd = {}
    cases = [(1, 64), (2, 64), (1, 16), (1, 320)]
    with torch.no_grad():
        for i, (B, T) in enumerate(cases):
            x, m = seeded_inputs(1000 + i, B, T)
            y = g(x, m)
            fwd["g_out_%dx%d" % (B, T)] = y.numpy()
            fwd["d_out_%dx%d" % (B, T)] = d(x).numpy()
            fwd["dg_out_%dx%d" % (B, T)] = d(y).numpy()
    meta["forward_cases"] = [{"B": B, "T": T, "seed": 1000 + i} for i, (B, T) in enumerate(cases)]
    meta["filler_seeds"] = {"G": 101, "D": 202}
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **fwd)

    # ---- 4. per-leaf activation digests at (1,64) ------------------------------------------
    digests = {}

    def hook(name):
        def fn(_m, _i, o):
            o = o.detach()
            digests.setdefault(name, []).append({
                "shape": list(o.shape), "mean": float(o.double().mean()), "std": float(o.double().std()),
                "absmax": float(o.abs().max()), "first": [float(v) for v in o.flatten()[:8]]})
        return fn

    hs = []
    for net, tag in ((g, "G"), (d, "D")):
        for name, mod_ in net.named_modules():
            if len(list(mod_.children())) == 0:
                hs.append(mod_.register_forward_hook(hook(tag + ":" + name)))
    x, m = seeded_inputs(1000, 1, 64)
    with torch.no_grad():
        d(g(x, m))
    for h in hs:
        h.remove()
    json.dump(digests, open(os.path.join(HERE, "layer_digests.json"), "w"), indent=0)

    # ---- 5. gradient goldens ----------------------------------------------------------------
    x, m = seeded_inputs(2000, 2, 64)
    x.requires_grad_(True)
    for p in list(g.parameters()) + list(d.parameters()):
        p.grad = None
    y = g(x, m)
    loss = torch.mean((1 - d(y)) ** 2) + 10.0 * torch.mean(torch.abs(x.detach() - y))
    loss.backward()
    grads = {"loss": np.float64(loss.item()), "dx": x.grad.numpy()}
    gnorm = {}
    for tag, net in (("G", g), ("D", d)):
        for n, p in net.named_parameters():
            if p.grad is None:
                gnorm[tag + ":" + n] = None
                continue
            gnorm[tag + ":" + n] = float(p.grad.double().norm())
            grads[tag + ":" + n] = p.grad.flatten()[:32].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "grads.npz"), **grads)
    meta["grad_case"] = {"B": 2, "T": 64, "seed": 2000, "loss": "mean((1-D(G(x,m)))^2)+10*mean|x-G(x,m)|"}
    json.dump(gnorm, open(os.path.join(HERE, "grad_norms.json"), "w"), indent=0)

    refs = (Generator, Discriminator, MaskCycleGANVCTraining, VCDataset, TrainLogger)
    for tag in TRAIN_CASES:
        run_train_case(tag, *refs)

    # ---- 7. dataset draws ---------------------------------------------------------------------
    np.random.seed(0)
    rs = np.random.RandomState(7)
    dsA = [rs.randn(80, 70 + 5 * i).astype(np.float32) for i in range(3)]
    dsB = [rs.randn(80, 90 + 3 * i).astype(np.float32) for i in range(4)]
    ds = VCDataset(dsA, dsB, n_frames=64, max_mask_len=25)
    draws = {}
    for k, idx in enumerate((0, 2, 1, 0)):
        a, ma, b, mb = ds[idx]
        draws["d%d_A" % k], draws["d%d_mA" % k], draws["d%d_B" % k], draws["d%d_mB" % k] = a, ma, b, mb
    np.savez_compressed(os.path.join(HERE, "dataset_draws.npz"), **draws)
    meta["dataset_case"] = {"np_seed": 0, "data_seed": 7, "indices": [0, 2, 1, 0], "len": len(ds)}

    json.dump(layout, open(os.path.join(HERE, "layout.json"), "w"), indent=0)
    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main(sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None)
