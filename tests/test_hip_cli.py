"""GPU end-to-end: python -m mask_cyclegan_vc.train (2 iterations + checkpoint), --continue_train resume,
python -m mask_cyclegan_vc.test, on a tiny synthetic VCC-shaped dataset written to disk."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_speaker(root, spk, seed, n):
    rs = np.random.RandomState(seed)
    d = os.path.join(root, spk)
    os.makedirs(d, exist_ok=True)
    mels = [rs.randn(80, 72 + 8 * i).astype(np.float32) for i in range(n)]
    with open(os.path.join(d, "%s_normalized.pickle" % spk), "wb") as fh:
        pickle.dump(mels, fh)
    np.savez(os.path.join(d, "%s_norm_stat.npz" % spk), mean=rs.randn(80, 1).astype(np.float32), std=(1 + rs.rand(80, 1)).astype(np.float32))
    return mels


def test_train_checkpoint_resume_and_inference(tmp_path):
    from mask_cyclegan_vc import test as test_cli
    from mask_cyclegan_vc import train as train_cli
    data = str(tmp_path / "data")
    _write_speaker(data, "SPKA", 1, 3)
    _write_speaker(data, "SPKB", 2, 3)
    common = ["--name", "run", "--save_dir", str(tmp_path / "res"), "--preprocessed_data_dir", data, "--speaker_A_id", "SPKA",
              "--speaker_B_id", "SPKB", "--batch_size", "2", "--num_epochs", "2", "--epochs_per_save", "1", "--max_mask_len", "25",
              "--steps_per_print", "2", "--seed", "0"]
    train_cli.main(common + ["--max_iters", "2", "--epochs_per_plot", "1"])   # epoch 1: batches of 2 and 1 (drop_last=False), validate, save
    # validation dump honoured by --epochs_per_plot (reference train.py:317-358 without the figure / vocoder back-ends)
    vdir = str(tmp_path / "res" / "run" / "validation")
    vnames = sorted(os.listdir(vdir))
    assert vnames == sorted("epoch00001_%s.npy" % k for k in ("real_A_spec", "real_B_spec", "fake_A_spec", "fake_B_spec", "real_speaker_A_mel",
                                                              "fake_speaker_A_mel", "real_speaker_B_mel", "fake_speaker_B_mel"))
    assert np.load(os.path.join(vdir, "epoch00001_fake_B_spec.npy")).shape == (80, 64)
    full = np.load(os.path.join(vdir, "epoch00001_fake_speaker_B_mel.npy"))
    assert full.shape == (80, 72) and np.isfinite(full).all()       # whole first utterance of speaker A (72 frames), converted
    ck = ck_dir = str(tmp_path / "res" / "run" / "ckpts")
    names = sorted(os.listdir(ck))
    assert names == ["00001_%s.pth.tar" % n for n in sorted(train_cli.NET_NAMES)]
    g = torch.load(os.path.join(ck, "00001_generator_A2B.pth.tar"), weights_only=False)
    assert len(g["model_state"]) == 114 and "convLayer.0.weight" in g["model_state"]
    assert sorted(g["optimizer"]["state"].keys()) == list(range(220))          # both generators' 110 tensors
    assert g["optimizer"]["param_groups"][0]["betas"] == (0.5, 0.999)
    d = torch.load(os.path.join(ck, "00001_discriminator_B2.pth.tar"), weights_only=False)
    dead = {n * 20 + i for n in range(4) for i in range(14, 18)}
    assert sorted(d["optimizer"]["state"].keys()) == sorted(set(range(80)) - dead)    # downSample4 has no Adam state
    assert float(d["optimizer"]["state"][0]["step"]) == 2.0
    # the saved generator state loads into a torch.optim.Adam over reference-shaped parameters
    ref_params = [torch.nn.Parameter(torch.zeros_like(v)) for k, v in g["model_state"].items() if not k.startswith("upSample2.")] * 2
    torch.optim.Adam(ref_params, lr=2e-4, betas=(0.5, 0.999)).load_state_dict(g["optimizer"])
    # the engine's optimizer export has the structure of a reference-written file (tests/golden/ckpt_structure.json)
    import json
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_structure.json")))
    for fname, key in (("00001_generator_A2B.pth.tar", "generator_A2B"), ("00001_discriminator_A.pth.tar", "discriminator_A")):
        ckf = torch.load(os.path.join(ck_dir, fname), weights_only=False)
        r = ref[key]
        assert list(ckf.keys()) == r["top_level_keys"] and list(ckf["model_state"].keys()) == r["model_state_order"]
        assert all(list(v.shape) == r["model_state"][k]["shape"] and str(v.dtype) == "torch." + r["model_state"][k]["tensor"] and v.device.type == "cpu"
                   for k, v in ckf["model_state"].items())
        st = ckf["optimizer"]["state"]
        assert sorted(st.keys()) == r["optimizer_state_indices"]
        assert {str(i): list(st[i]["exp_avg"].shape) for i in sorted(st)} == r["optimizer_state_shapes"]
        e0 = st[sorted(st)[0]]
        assert sorted(e0.keys()) == sorted(r["optimizer_state_entry"].keys())
        assert all(e0[k].dtype == torch.float32 and e0[k].device.type == "cpu" and list(e0[k].shape) == r["optimizer_state_entry"][k]["shape"] for k in e0)
        grp = ckf["optimizer"]["param_groups"][0]
        assert sorted(grp.keys()) == sorted(r["param_group"].keys())
        for k, d in r["param_group"].items():
            assert type(grp[k]).__name__ == d["type"], k
            if k == "params":
                assert grp[k] == list(range(d["len"]))
            elif k not in ("lr", "betas"):
                assert grp[k] == d["value"], k
    # resume: epoch 2 starts from the epoch-1 files; this time on the reference's RNG-exact host sampler path
    train_cli.main(common + ["--continue_train", "--max_iters", "1", "--host_sampler"])
    log = open(str(tmp_path / "res" / "run" / "run.log")).read()
    assert "[start of epoch 2]" in log and "g_loss" in log
    # inference CLI
    outs = test_cli.MaskCycleGANVCTesting(test_cli.CycleGANTestArgParser().parse_args(
        ["--name", "run", "--save_dir", str(tmp_path / "res"), "--preprocessed_data_dir", data, "--speaker_A_id", "SPKA",
         "--speaker_B_id", "SPKB", "--ckpt_dir", ck, "--load_epoch", "1", "--model_name", "generator_A2B"])).test()
    assert len(outs) == 3
    m = np.load(outs[1])
    assert m.shape == (80, 80) and np.isfinite(m).all()            # T=80 utterance keeps its length (multiple of 4)
