"""CPU checks of the drop-in CLI surface: flags/defaults, dataset RNG parity, checkpoint layout, logger counters."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from args import CycleGANTestArgParser, CycleGANTrainArgParser
from dataset.vc_dataset import VCDataset
from logger.train_logger import TrainLogger
from saver.model_saver import ModelSaver


def test_train_flags_and_defaults_match_reference(tmp_path):
    # effective defaults of reference args/{base,train,cycleGAN_train}_arg_parser.py incl. its set_defaults() (:51)
    expect = dict(name="debug", batch_size=1, seed=0, steps_per_print=100, epochs_per_save=1, start_epoch=1, load_epoch=0,
                  num_epochs=50, decay_after=1e4, stop_identity_after=1e4, max_ckpts=3, continue_train=False, sample_rate=22050,
                  speaker_A_id="28", speaker_B_id="DCB_se2_ag3_m_02_1", preprocessed_data_dir="vcc2018_training_preprocessed/",
                  generator_lr=2e-4, discriminator_lr=1e-4, cycle_loss_lambda=10, identity_loss_lambda=5, epochs_per_plot=2,
                  num_frames=64, num_frames_validation=320, max_mask_len=32)
    args = CycleGANTrainArgParser().parse_args(["--save_dir", str(tmp_path)])
    for k, v in expect.items():
        assert getattr(args, k) == v, k
    assert args.isTrain is True and args.ckpt_dir == os.path.join(str(tmp_path), "debug", "ckpts") and os.path.isdir(args.ckpt_dir)
    assert os.path.exists(os.path.join(str(tmp_path), "debug", "train_args.json"))
    assert args.device in ("cuda", "cpu")
    # the canonical command line of bash_scripts/mask_cyclegan_train.sh parses
    args = CycleGANTrainArgParser().parse_args(
        "--name n --seed 0 --save_dir {d} --preprocessed_data_dir x --speaker_A_id VCC2SF3 --speaker_B_id VCC2TF1 --epochs_per_save 100 "
        "--epochs_per_plot 10 --num_epochs 6172 --decay_after 2e5 --stop_identity_after 1e4 --batch_size 1 --sample_rate 22050 "
        "--num_frames 64 --max_mask_len 25 --gpu_ids 0".format(d=tmp_path).split())
    assert args.num_epochs == 6172 and args.decay_after == 2e5 and args.max_mask_len == 25
    # README's --lr is NOT a flag of the reference either
    with pytest.raises(SystemExit):
        CycleGANTrainArgParser().parse_args(["--save_dir", str(tmp_path), "--lr", "5e-4"])


def test_resume_epoch_resolution(tmp_path):
    d = tmp_path / "r" / "ckpts"
    d.mkdir(parents=True)
    for e in (100, 200):
        (d / ("%05d_generator_A2B.pth.tar" % e)).write_bytes(b"")
    a = CycleGANTrainArgParser().parse_args(["--save_dir", str(tmp_path), "--name", "r", "--continue_train"])
    assert (a.load_epoch, a.start_epoch) == (200, 201)           # last saved file by sorted name
    a = CycleGANTrainArgParser().parse_args(["--save_dir", str(tmp_path), "--name", "r", "--continue_train", "--load_epoch", "100"])
    assert (a.load_epoch, a.start_epoch) == (100, 101)
    a = CycleGANTrainArgParser().parse_args(["--save_dir", str(tmp_path), "--name", "r", "--continue_train", "--start_epoch", "7"])
    assert (a.load_epoch, a.start_epoch) == (6, 7)
    t = CycleGANTestArgParser().parse_args(["--save_dir", str(tmp_path), "--name", "r", "--ckpt_dir", str(d), "--load_epoch", "200"])
    assert t.isTrain is False and t.model_name == "generator_A2B" and t.start_epoch == 201


def test_dataset_draws_are_bit_identical_to_reference(golden_dir):
    gold = np.load(os.path.join(golden_dir, "dataset_draws.npz"))
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["dataset_case"]
    np.random.seed(meta["np_seed"])
    rs = np.random.RandomState(meta["data_seed"])
    dsA = [rs.randn(80, 70 + 5 * i).astype(np.float32) for i in range(3)]
    dsB = [rs.randn(80, 90 + 3 * i).astype(np.float32) for i in range(4)]
    ds = VCDataset(dsA, dsB, n_frames=64, max_mask_len=25)
    assert len(ds) == meta["len"]
    for k, idx in enumerate(meta["indices"]):
        a, ma, b, mb = ds[idx]
        for mine, key in ((a, "A"), (ma, "mA"), (b, "B"), (mb, "mB")):
            assert np.array_equal(mine, gold["d%d_%s" % (k, key)]), (k, key)
    # validation mode returns whole utterances
    v = VCDataset(dsA, dsB, valid=True)
    assert v[1][0] is dsA[1] and v[1][1] is dsB[1]


def test_device_sampler_restatement_follows_the_reference_distributions():
    """oracle/sampler_oracle.py restates the on-device sampler's counter-based draws (the HIP kernel is checked bit-exactly
    against it on the GPU, tests/test_hip_sampler.py).  Here: the DISTRIBUTIONS are the reference's (vc_dataset.py:33-70):
    utterance uniform with replacement, crop start uniform on {0..len-T}, mask size uniform on {0..max_mask_len-1}, mask
    start uniform on {0..T-size-1}; chi-square against those laws on 40k draws (99.9 % quantiles)."""
    import sampler_oracle as so
    from scipy import stats
    lens_a, lens_b = [70, 75, 80, 64], [90, 93, 96, 99, 102]
    T, mml, B, steps = 64, 25, 8, 2500
    idx = np.concatenate([so.draw_indices(lens_a, lens_b, B, T, mml, seed=11, step=s) for s in range(steps)])   # [steps*B, 2, 4]
    n = idx.shape[0]
    for side, lens in enumerate((lens_a, lens_b)):
        utt, lo, size, start = (idx[:, side, k] for k in range(4))
        assert utt.min() >= 0 and utt.max() == len(lens) - 1
        assert stats.chisquare(np.bincount(utt, minlength=len(lens))).pvalue > 1e-3
        assert stats.chisquare(np.bincount(size, minlength=mml)).pvalue > 1e-3 and size.max() == mml - 1
        for u, ln in enumerate(lens):                      # crop start | utterance ~ U{0..len-T}
            sel = lo[utt == u]
            assert sel.min() >= 0 and sel.max() <= ln - T
            if ln > T:
                assert stats.chisquare(np.bincount(sel, minlength=ln - T + 1)).pvalue > 1e-3
        for sz in (0, 7, 24):                               # mask start | size ~ U{0..T-size-1}
            sel = start[size == sz]
            assert sel.max() <= T - sz - 1
            assert stats.chisquare(np.bincount(sel, minlength=T - sz)).pvalue > 1e-3
        assert (start + size <= T).all()
    # the two speakers and consecutive steps are independent streams: no repeated rows
    assert len({tuple(r) for r in idx.reshape(n, 8)}) > 0.99 * n
    # materialised batch: crops are slices of the chosen utterances, masks are per-frame (identical over the 80 bins)
    rs = np.random.RandomState(5)
    dA = [rs.randn(80, ln).astype(np.float32) for ln in lens_a]
    dB = [rs.randn(80, ln).astype(np.float32) for ln in lens_b]
    (xa, ma, xb, mb), ix = so.draw_batch(dA, dB, 5, T, mml, seed=3, step=9)
    assert xa.shape == ma.shape == xb.shape == mb.shape == (5, 80, 64)
    assert set(np.unique(ma)).issubset({0.0, 1.0}) and (ma == ma[:, :1, :]).all() and (mb == mb[:, :1, :]).all()
    u, lo, size, start = ix[2, 1]
    assert np.array_equal(xb[2], dB[u][:, lo:lo + T]) and mb[2, 0].sum() == T - size and (mb[2, 0, start:start + size] == 0).all()


def test_preprocessed_format_writer_roundtrip(tmp_path):
    """data_preprocessing.preprocess_vcc2018 writes what train.load_speaker (and the reference trainer, train.py:51-64)
    reads; normalisation per reference preprocess_vcc2018.py:35-47 (std + 1e-9, utterances under 64 frames dropped)."""
    from data_preprocessing.preprocess_vcc2018 import main as prep_main, normalize_mels
    import pickle
    rs = np.random.RandomState(0)
    mels = [(3.0 + 2.0 * rs.randn(80, n)).astype(np.float32) for n in (70, 40, 128, 64)]
    for i, m in enumerate(mels):
        os.makedirs(tmp_path / "mels" / "SPK", exist_ok=True)
        np.save(tmp_path / "mels" / "SPK" / ("u%d.npy" % i), m)
    prep_main(["--mel_directory", str(tmp_path / "mels"), "--preprocessed_data_directory", str(tmp_path / "out"), "--speaker_ids", "SPK"])
    with open(tmp_path / "out" / "SPK" / "SPK_normalized.pickle", "rb") as fh:
        norm = pickle.load(fh)
    stat = np.load(tmp_path / "out" / "SPK" / "SPK_norm_stat.npz")
    kept = [m for m in mels if m.shape[1] >= 64]
    assert len(norm) == 3 and [n.shape for n in norm] == [m.shape for m in kept]
    cat = np.concatenate(kept, axis=1)
    assert stat["mean"].shape == stat["std"].shape == (80, 1)
    assert np.allclose(stat["mean"], cat.mean(1, keepdims=True)) and np.allclose(stat["std"], cat.std(1, keepdims=True) + 1e-9)
    assert all(n.dtype == np.float32 for n in norm)
    assert np.allclose(norm[1], (kept[1] - stat["mean"]) / stat["std"], atol=1e-6)
    again, _, _ = normalize_mels(kept)
    assert all(np.array_equal(a, b) for a, b in zip(again, norm))


class _FakeAdam(object):
    def __init__(self):
        self.sd = {"state": {0: {"step": torch.tensor(3.0), "exp_avg": torch.ones(2), "exp_avg_sq": torch.ones(2)}},
                   "param_groups": [{"lr": 2e-4, "betas": (0.5, 0.999), "params": [0]}]}

    def state_dict(self):
        return self.sd

    def load_state_dict(self, sd):
        self.sd = sd


def test_checkpoint_layout_roundtrip(tmp_path):
    from mask_cyclegan_vc.model import Discriminator
    from argparse import Namespace
    args = Namespace(ckpt_dir=str(tmp_path), load_epoch=7, gpu_ids=[0])
    saver = ModelSaver(args)
    d = Discriminator()
    path = saver.save(7, d, _FakeAdam(), None, "cpu", "discriminator_A")
    assert os.path.basename(path) == "00007_discriminator_A.pth.tar"
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"ckpt_info", "model_class", "model_state", "optimizer", "lr_scheduler"}
    assert ck["ckpt_info"] == {"epoch": 7} and ck["model_class"] == "Discriminator" and ck["lr_scheduler"] is None
    assert list(ck["model_state"]) == list(d.state_dict()) and len(ck["model_state"]) == 20
    assert all(v.device.type == "cpu" for v in ck["model_state"].values())
    d2 = Discriminator()
    opt = _FakeAdam(); opt.sd = None
    saver.load_model(d2, "discriminator_A", None, opt)
    assert all(torch.equal(a, b) for a, b in zip(d.state_dict().values(), d2.state_dict().values()))
    assert opt.sd["param_groups"][0]["lr"] == 2e-4


def _describe(v):
    if torch.is_tensor(v):
        return {"tensor": str(v.dtype).replace("torch.", ""), "shape": list(v.shape), "device": v.device.type}
    if isinstance(v, (list, tuple)):
        return {"type": type(v).__name__, "len": len(v), "elem": type(v[0]).__name__ if len(v) else None}
    return {"type": type(v).__name__, "value": v if isinstance(v, (int, float, bool, str, type(None))) else None}


def test_saver_writes_the_structure_of_a_reference_written_checkpoint(tmp_path, golden_dir):
    """tests/golden/ckpt_structure.json describes files written by the REFERENCE's own ModelSaver.save (make_golden_ckpt.py):
    file name, top-level keys, model_state keys / dtypes / shapes, optimizer state indices and entry layout, param_groups keys.
    This repo's saver + modules + a torch.optim.Adam must produce the same structure."""
    from argparse import Namespace
    from mask_cyclegan_vc.model import Discriminator, Generator
    ref = json.load(open(os.path.join(golden_dir, "ckpt_structure.json")))
    plain = json.load(open(os.path.join(golden_dir, "step_plain.json")))
    saver = ModelSaver(Namespace(ckpt_dir=str(tmp_path), load_epoch=3, gpu_ids=[0]))
    torch.manual_seed(0)
    for name, model in (("generator_A2B", Generator()), ("discriminator_A", Discriminator())):
        n_copies = 2 if name.startswith("generator") else 4               # the optimizers span both G / all four D (train.py:113-122)
        others = [type(model)() for _ in range(n_copies - 1)]
        params = [p for m in [model] + others for p in m.parameters()]
        opt = torch.optim.Adam(params, lr=2e-4 if n_copies == 2 else 1e-4, betas=(0.5, 0.999))
        for m in [model] + others:
            for pn, p in m.named_parameters():
                if not pn.startswith("downSample4."):
                    p.grad = torch.ones_like(p)
        opt.step()
        path = saver.save(3, model, opt, None, "cpu", name)
        r = ref[name]
        assert os.path.basename(path) == r["file_name"]
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert list(ck.keys()) == r["top_level_keys"]
        assert ck["ckpt_info"] == r["ckpt_info"] and ck["model_class"] == r["model_class"] and ck["lr_scheduler"] is r["lr_scheduler"]
        assert type(ck["model_state"]).__name__ == r["model_state_type"]
        assert {k: _describe(v) for k, v in ck["model_state"].items()} == r["model_state"]
        assert list(ck["model_state"].keys()) == r["model_state_order"]
        st = ck["optimizer"]["state"]
        assert list(ck["optimizer"].keys()) == r["optimizer_keys"] and sorted(st.keys()) == r["optimizer_state_indices"]
        assert {k: _describe(v) for k, v in st[sorted(st)[0]].items()} == r["optimizer_state_entry"]
        assert {str(i): list(st[i]["exp_avg"].shape) for i in sorted(st)} == r["optimizer_state_shapes"]
        grp = ck["optimizer"]["param_groups"][0]
        want = dict(r["param_group"])
        if n_copies == 4:
            want["lr"] = dict(want["lr"], value=1e-4)
        assert {k: _describe(v) for k, v in grp.items()} == want
        assert sorted(grp.keys()) == plain["adam_group_keys"]            # recorded from the unmodified train() run as well


def test_logger_counters(tmp_path):
    from argparse import Namespace
    os.makedirs(tmp_path / "n")
    lg = TrainLogger(Namespace(batch_size=2, save_dir=str(tmp_path), name="n", start_epoch=3, steps_per_print=2, num_epochs=5), 81)
    assert lg.global_step == 162 and lg.epoch == 3          # round_down((3-1)*81, 2)
    lg.start_epoch(); lg.start_iter(); lg.log_iter({"g_loss": 1.0, "d_loss": 0.5}); lg.end_iter()
    assert (lg.iter, lg.global_step) == (2, 164)
    lg.end_epoch()
    assert lg.epoch == 4 and not lg.is_finished_training()
    text = open(tmp_path / "n" / "n.log").read()
    assert "[start of epoch 3]" in text and "g_loss: 1.000" in text


REF_ROOT = "/root/reference"


def _load_reference_module(name, rel):
    """The unmodified reference file as a private module (build container only; nothing of it is copied or shipped)."""
    import importlib.util
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.isdir(REF_ROOT), reason="needs the reference checkout (build container only)")
def test_checkpoints_interchange_with_the_reference_saver(tmp_path):
    """SURVEY.md section 8 f1, both directions, with the REFERENCE's own code on the other side: a file written by the reference's
    ``ModelSaver.save`` (saver/model_saver.py:46-93) from reference modules + torch.optim.Adam loads through this repo's
    ``ModelSaver.load_model`` into this repo's modules (strict key set, bit-equal tensors, optimizer state), and a file written
    by this repo's saver loads through the reference's ``load_model`` (:95-123) into reference modules."""
    from argparse import Namespace
    from mask_cyclegan_vc.model import Discriminator, Generator
    ref_model = _load_reference_module("_ref_model_for_ckpt_test", "mask_cyclegan_vc/model.py")
    ref_saver_mod = _load_reference_module("_ref_saver_for_ckpt_test", "saver/model_saver.py")
    d_ref, d_ours = tmp_path / "ref", tmp_path / "ours"
    d_ref.mkdir(); d_ours.mkdir()
    ref_saver = ref_saver_mod.ModelSaver(Namespace(ckpt_dir=str(d_ref), load_epoch=5, gpu_ids=["cpu"]))
    our_saver = ModelSaver(Namespace(ckpt_dir=str(d_ref), load_epoch=5, gpu_ids=["cpu"]))

    def adam_with_state(model, lr, dead_prefix=None):
        opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.5, 0.999))
        g = torch.Generator().manual_seed(11)
        for n, p in model.named_parameters():
            if dead_prefix is None or not n.startswith(dead_prefix):
                p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        return opt

    def same_opt(a, b):
        sa, sb = a.state_dict(), b.state_dict()
        assert sorted(sa["state"]) == sorted(sb["state"])
        for k in sa["state"]:
            for f in ("step", "exp_avg", "exp_avg_sq"):
                assert torch.equal(torch.as_tensor(sa["state"][k][f]), torch.as_tensor(sb["state"][k][f])), (k, f)
        ga, gb = sa["param_groups"][0], sb["param_groups"][0]
        assert ga["lr"] == gb["lr"] and tuple(ga["betas"]) == tuple(gb["betas"]) and ga["params"] == gb["params"]

    for name, RefCls, OurCls, lr, dead in (("generator_A2B", ref_model.Generator, Generator, 2e-4, None),
                                           ("discriminator_A", ref_model.Discriminator, Discriminator, 1e-4, "downSample4.")):
        # ---- reference writes, we read
        torch.manual_seed(1)
        src = RefCls()
        opt_src = adam_with_state(src, lr, dead)
        ref_saver.save(5, src, opt_src, None, "cpu", name)
        torch.manual_seed(2)
        dst = OurCls()
        opt_dst = torch.optim.Adam(dst.parameters(), lr=9.0, betas=(0.9, 0.9))
        ck = our_saver.load_model(dst, name, None, opt_dst)
        assert ck["ckpt_info"] == {"epoch": 5} and ck["model_class"] == RefCls.__name__
        assert list(dst.state_dict()) == list(src.state_dict())
        assert all(torch.equal(a, b) for a, b in zip(dst.state_dict().values(), src.state_dict().values()))
        same_opt(opt_dst, opt_src)
        # ---- we write, the reference reads
        torch.manual_seed(3)
        mine = OurCls()
        opt_mine = adam_with_state(mine, lr, dead)
        ModelSaver(Namespace(ckpt_dir=str(d_ours), load_epoch=6, gpu_ids=["cpu"])).save(6, mine, opt_mine, None, "cpu", name)
        torch.manual_seed(4)
        theirs = RefCls()
        opt_theirs = torch.optim.Adam(theirs.parameters(), lr=9.0, betas=(0.9, 0.9))
        ref_saver_mod.ModelSaver(Namespace(ckpt_dir=str(d_ours), load_epoch=6, gpu_ids=["cpu"])).load_model(theirs, name, None, opt_theirs)
        assert list(theirs.state_dict()) == list(mine.state_dict())
        assert all(torch.equal(a, b) for a, b in zip(theirs.state_dict().values(), mine.state_dict().values()))
        same_opt(opt_theirs, opt_mine)
