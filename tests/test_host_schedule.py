"""Host-side step bookkeeping (mask_cyclegan_vc/schedule.py) against the reference's unmodified train() trace:
global_step advances by batch size, the LR-decay call-site bug, identity-loss cut-off."""
import json
import os

from mask_cyclegan_vc.schedule import StepSchedule


def test_schedule_replays_reference_decay_trace(golden_dir):
    js = json.load(open(os.path.join(golden_dir, "step_decay.json")))
    cfg = js["config"]
    s = StepSchedule(generator_lr=cfg["g_lr"], discriminator_lr=cfg["d_lr"], num_epochs=cfg["num_epochs"], n_samples=cfg["n_utt"],
                     batch_size=cfg["batch_size"], decay_after=cfg["decay_after"], stop_identity_after=cfg["stop_identity_after"])
    for tr in js["trace"]:
        # the fixture is taken inside logger.end_iter(), before this iteration's lr adjustment
        assert abs(s.g_opt_lr - tr["g_opt_lr"]) < 1e-15 and abs(s.d_opt_lr - tr["d_opt_lr"]) < 1e-15
        assert s.identity_loss_lambda == tr["identity_lambda_before_check"]
        s.end_iteration()
        assert s.global_step == tr["global_step"]
    fin = js["final"]
    assert abs(s.g_opt_lr - fin["g_opt_lr"]) < 1e-15            # generator optimizer holds the decayed DISCRIMINATOR lr
    assert abs(s.d_opt_lr - fin["d_opt_lr"]) < 1e-15            # discriminator optimizer never decays
    assert abs(s.generator_lr - fin["generator_lr_attr"]) < 1e-15
    assert abs(s.discriminator_lr - fin["discriminator_lr_attr"]) < 1e-15
    assert s.identity_loss_lambda == fin["identity_loss_lambda"] == 0
    assert fin["g_opt_lr"] != cfg["g_lr"] and fin["d_opt_lr"] == cfg["d_lr"]


def test_schedule_resume_and_world_size():
    s1 = StepSchedule(n_samples=81, batch_size=1, start_epoch=3, dataset_len=81, world_size=1)
    assert s1.global_step == 162                                 # base_logger.py:55-56
    s = StepSchedule(n_samples=81, batch_size=1, start_epoch=3, dataset_len=81, world_size=8)
    assert s.global_step == 162 * 8                              # an epoch advances global_step by dataset_len * world
    s.end_iteration()
    assert s.global_step == 162 * 8 + 8                          # every rank consumed batch_size samples


def test_resumed_global_step_equals_the_uninterrupted_counter_under_data_parallelism():
    """with R ranks the resumed counter used to be R times too small (identity loss switched back on, LR decay
    start shifted).  Two epochs of 81 single-sample iterations on 2 ranks, then a resume at epoch 3."""
    from argparse import Namespace
    import os
    import tempfile
    from logger.train_logger import TrainLogger
    run = StepSchedule(n_samples=81, batch_size=1, start_epoch=1, dataset_len=81, world_size=2)
    for _ in range(2 * 81):
        run.end_iteration()
    resumed = StepSchedule(n_samples=81, batch_size=1, start_epoch=3, dataset_len=81, world_size=2)
    assert resumed.global_step == run.global_step == 324
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "n"))
        lg = TrainLogger(Namespace(batch_size=1, save_dir=d, name="n", start_epoch=3, steps_per_print=1, num_epochs=5), 81, world_size=2)
        assert lg.global_step == run.global_step
