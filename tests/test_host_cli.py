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
    # O(batch) sampler: same mask law
    xa, ma, xb, mb = ds.draw_batch(5, np.random.RandomState(3))
    assert xa.shape == ma.shape == xb.shape == mb.shape == (5, 80, 64)
    assert set(np.unique(ma)).issubset({0.0, 1.0}) and (ma == ma[:, :1, :]).all()


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
