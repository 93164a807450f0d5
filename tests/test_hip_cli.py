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


def _args(tmp_path, data, name, extra):
    from args.cycleGAN_train_arg_parser import CycleGANTrainArgParser
    return CycleGANTrainArgParser().parse_args(
        ["--name", name, "--save_dir", str(tmp_path / "res"), "--preprocessed_data_dir", data, "--speaker_A_id", "SPKA", "--speaker_B_id", "SPKB",
         "--batch_size", "2", "--epochs_per_save", "1", "--max_mask_len", "25", "--steps_per_print", "1", "--seed", "0"] + extra)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_validation_dump_and_converted_mels_match_the_oracle(tmp_path):
    """SURVEY.md section 8 f4 by VALUE (reference train.py:317-358, test.py:85-119): the spectrograms the trainer dumps every
    ``--epochs_per_plot`` epochs and the files ``mask_cyclegan_vc.test`` writes through its bucketed two-stream driver are compared with
    ``oracle.generator_forward`` run on the same utterances with the weights of the checkpoint saved right after -- fp32 <= 1e-3,
    bf16 <= 2e-2 relative (in the normalised domain the network works in)."""
    import mcvc_oracle as orc
    from mask_cyclegan_vc import test as test_cli
    from mask_cyclegan_vc import train as train_cli
    data = str(tmp_path / "data")
    mels_A = _write_speaker(data, "SPKA", 1, 5)          # lengths 72, 80, 88, 96, 104
    mels_B = _write_speaker(data, "SPKB", 2, 5)
    mels_A.append(mels_A[1] * 0.5); mels_B.append(mels_B[1] * 0.5)          # a second 80-frame utterance: one bucket of two (batched forward)
    for spk, mels in (("SPKA", mels_A), ("SPKB", mels_B)):
        with open(os.path.join(data, spk, "%s_normalized.pickle" % spk), "wb") as fh:
            pickle.dump(mels, fh)
    job = train_cli.MaskCycleGANVCTraining(_args(tmp_path, data, "val", ["--num_epochs", "1", "--max_iters", "2", "--epochs_per_plot", "1"]))
    try:
        job.train()
        real_A, mask_A, real_B, mask_B = [t.detach().cpu() for t in job.engine.static_in]       # the last training minibatch
    finally:
        job.close()
    vdir = str(tmp_path / "res" / "val" / "validation")
    ck = str(tmp_path / "res" / "val" / "ckpts")
    gA2B = torch.load(os.path.join(ck, "00001_generator_A2B.pth.tar"), weights_only=False)["model_state"]
    gB2A = torch.load(os.path.join(ck, "00001_generator_B2A.pth.tar"), weights_only=False)["model_state"]
    stat = {s: np.load(os.path.join(data, s, "%s_norm_stat.npz" % s)) for s in ("SPKA", "SPKB")}
    load = lambda k: np.load(os.path.join(vdir, "epoch00001_%s.npy" % k))      # noqa: E731
    with torch.no_grad():
        # the last training pair (train.py:321-333): generated_B = G_A2B(real_A, mask_A), generated_A = G_B2A(real_B, mask_B), first sample
        fake_B = orc.generator_forward(gA2B, real_A[:1], mask_A[:1])[0].numpy()
        fake_A = orc.generator_forward(gB2A, real_B[:1], mask_B[:1])[0].numpy()
        assert np.array_equal(load("real_A_spec"), real_A[0].numpy()) and np.array_equal(load("real_B_spec"), real_B[0].numpy())
        assert _rel(load("fake_B_spec"), fake_B) < 1e-3 and _rel(load("fake_A_spec"), fake_A) < 1e-3
        # whole first validation utterances, all-ones mask (train.py:336-358)
        xa, xb = torch.from_numpy(mels_A[0])[None], torch.from_numpy(mels_B[0])[None]
        full_B = orc.generator_forward(gA2B, xa, torch.ones_like(xa))[0].numpy()
        full_A = orc.generator_forward(gB2A, xb, torch.ones_like(xb))[0].numpy()
    norm = lambda m, s: (m - stat[s]["mean"]) / stat[s]["std"]                  # noqa: E731
    assert _rel(norm(load("fake_speaker_B_mel"), "SPKB"), full_B) < 1e-3       # de-normalised with the TARGET speaker's statistics
    assert _rel(norm(load("fake_speaker_A_mel"), "SPKA"), full_A) < 1e-3       # (reference train.py:343-350)
    assert _rel(load("real_speaker_A_mel"), mels_A[0] * stat["SPKA"]["std"] + stat["SPKA"]["mean"]) < 1e-6
    # ---- inference CLI through the bucketed two-stream driver, fp32 and bf16: every utterance, incl. the batched bucket
    for dtype, tol in (("f32", 1e-3), ("bf16", 2e-2)):
        targs = test_cli.CycleGANTestArgParser().parse_args(
            ["--name", "val_" + dtype, "--save_dir", str(tmp_path / "res"), "--preprocessed_data_dir", data, "--speaker_A_id", "SPKA",
             "--speaker_B_id", "SPKB", "--ckpt_dir", ck, "--load_epoch", "1", "--model_name", "generator_A2B", "--dtype", dtype])
        outs = test_cli.MaskCycleGANVCTesting(targs).test()
        assert len(outs) == len(mels_A)
        worst = 0.0
        for i, path in enumerate(outs):
            x = torch.from_numpy(mels_A[i])[None]
            with torch.no_grad():
                ref = orc.generator_forward(gA2B, x, torch.ones_like(x))[0].numpy()
            got = norm(np.load(path), "SPKB")                                  # test.py de-normalises with the TARGET speaker's statistics
            assert got.shape == ref.shape
            worst = max(worst, _rel(got, ref))
        print("converted mels vs oracle (%s): worst rel-L2 %.3e" % (dtype, worst))
        assert worst < tol, (dtype, worst)


def test_resume_is_equivalent_to_an_uninterrupted_run(tmp_path):
    """Reference train.py:125-137 + logger/base_logger.py:55-56: in bit-reproducible mode, two epochs straight == one epoch, process
    exit, ``--continue_train`` for the second -- parameters, Adam moments and step counts of all six networks bit-equal, same
    ``global_step``, same learning rates.  (Holds before ``decay_after`` only, in the reference as here: the trainer's python-float
    learning rates restart from the flag values on resume -- train.py:35-36 -- while the optimizers keep the checkpointed ones.)"""
    from mask_cyclegan_vc import _hip
    from mask_cyclegan_vc import train as train_cli
    data = str(tmp_path / "data")
    _write_speaker(data, "SPKA", 1, 4)
    _write_speaker(data, "SPKB", 2, 4)
    was = _hip.lib().mcvc_set_deterministic(1)
    try:
        def run(name, extra):
            job = train_cli.MaskCycleGANVCTraining(_args(tmp_path, data, name, ["--num_epochs", "2", "--epochs_per_plot", "0"] + extra))
            try:
                job.train()
                return dict(global_step=job.engine.sched.global_step, logger_step=job.logger.global_step, g_lr=job.engine.sched.g_opt_lr,
                            d_lr=job.engine.sched.d_opt_lr, g_step=job.engine.g_group.step, d_step=job.engine.d_group.step,
                            ident=job.engine.sched.identity_loss_lambda, sampler_step=job.sampler.step)
            finally:
                job.close()
        straight = run("straight", [])
        first = run("resumed", ["--max_iters", "2"])                    # epoch 1 = two iterations of batch 2 over the 4 utterances, then exit
        assert first["g_step"] == 2 and first["global_step"] == 4
        resumed = run("resumed", ["--continue_train"])                  # picks up 00001_*, runs epoch 2
        assert resumed == straight, (resumed, straight)
        assert straight["g_step"] == 4 and straight["global_step"] == 8 and straight["sampler_step"] == 4
        for n in train_cli.NET_NAMES:
            a = torch.load(str(tmp_path / "res" / "straight" / "ckpts" / ("00002_%s.pth.tar" % n)), weights_only=False)
            b = torch.load(str(tmp_path / "res" / "resumed" / "ckpts" / ("00002_%s.pth.tar" % n)), weights_only=False)
            assert list(a["model_state"]) == list(b["model_state"])
            for k in a["model_state"]:
                assert torch.equal(a["model_state"][k], b["model_state"][k]), (n, k)
            sa, sb = a["optimizer"], b["optimizer"]
            assert sorted(sa["state"]) == sorted(sb["state"]) and sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"]
            for i in sa["state"]:
                assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"]) == 4.0
                assert torch.equal(sa["state"][i]["exp_avg"], sb["state"][i]["exp_avg"]), (n, i)
                assert torch.equal(sa["state"][i]["exp_avg_sq"], sb["state"][i]["exp_avg_sq"]), (n, i)
    finally:
        _hip.lib().mcvc_set_deterministic(was)
