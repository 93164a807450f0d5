"""GPU: the on-device input pipeline (csrc/sampler_kernels.hip via mcvc_draw_batch / dataset.device_sampler.DeviceSampler)
against its CPU restatement oracle/sampler_oracle.py -- index work, so the bar is BIT-EXACT -- plus the distribution laws
of the reference's dataset (vc_dataset.py:33-70) on draws made by the kernel itself, and the engine fed by the sampler."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sampler_oracle as so  # noqa: E402
from dataset.device_sampler import DeviceSampler  # noqa: E402


def _data(seed, lens):
    rs = np.random.RandomState(seed)
    return [rs.randn(80, n).astype(np.float32) for n in lens]


@pytest.mark.parametrize("B,T,mml", [(1, 64, 25), (8, 64, 25), (32, 64, 32), (3, 48, 1), (2, 64, 64)])
def test_draws_match_the_restatement_bit_exactly(B, T, mml):
    dA, dB = _data(1, [64, 70, 99, 333, 64, 128]), _data(2, [65, 64, 200, 77])        # ragged, incl. exactly-T utterances
    sm = DeviceSampler(dA, dB, n_frames=T, max_mask_len=mml, seed=1234)
    bufs = [torch.empty(B, 80, T, device="cuda") for _ in range(4)]
    draws = torch.zeros(B * 8, dtype=torch.int32, device="cuda")
    for step in (0, 1, 7, 2 ** 33 + 5):
        sm.draw_into(*bufs, step=step, draws=draws)
        ref, idx = so.draw_batch(dA, dB, B, T, mml, seed=1234, step=step)
        assert np.array_equal(draws.cpu().numpy().reshape(B, 2, 4), idx)
        for got, want in zip(bufs, ref):
            assert np.array_equal(got.cpu().numpy(), want)
    # the running counter advances one minibatch per call and reproduces the explicit-step draws
    sm.step = 7
    sm.draw_into(*bufs)
    assert sm.step == 8
    assert np.array_equal(bufs[0].cpu().numpy(), so.draw_batch(dA, dB, B, T, mml, 1234, 7)[0][0])


def test_device_draws_follow_the_reference_distributions():
    from scipy import stats
    lens_a, lens_b = [70, 75, 80, 64], [90, 93, 96, 99, 102]
    T, mml, B = 64, 25, 32
    sm = DeviceSampler(_data(3, lens_a), _data(4, lens_b), n_frames=T, max_mask_len=mml, seed=99)
    bufs = [torch.empty(B, 80, T, device="cuda") for _ in range(4)]
    draws = torch.zeros(B * 8, dtype=torch.int32, device="cuda")
    rows = []
    for _ in range(800):
        sm.draw_into(*bufs, draws=draws)
        rows.append(draws.cpu().numpy().reshape(B, 2, 4).copy())
    idx = np.concatenate(rows)
    for side, lens in enumerate((lens_a, lens_b)):
        utt, lo, size, start = (idx[:, side, k] for k in range(4))
        assert stats.chisquare(np.bincount(utt, minlength=len(lens))).pvalue > 1e-3
        assert stats.chisquare(np.bincount(size, minlength=mml)).pvalue > 1e-3
        for u, ln in enumerate(lens):
            sel = lo[utt == u]
            assert sel.min() >= 0 and sel.max() <= ln - T
            if ln > T:
                assert stats.chisquare(np.bincount(sel, minlength=ln - T + 1)).pvalue > 1e-3
        sel = start[size == 10]
        assert sel.max() <= T - 11 and stats.chisquare(np.bincount(sel, minlength=T - 10)).pvalue > 1e-3
    # last minibatch on the device: masks are {0,1}, constant over the 80 bins, zero exactly on [start, start+size)
    m = bufs[1].cpu().numpy()
    last = idx[-B:, 0]
    assert set(np.unique(m)).issubset({0.0, 1.0}) and (m == m[:, :1, :]).all()
    for b in range(B):
        assert m[b, 0].sum() == T - last[b, 2] and (m[b, 0, last[b, 3]:last[b, 3] + last[b, 2]] == 0).all()


def test_sampler_rejects_bad_input():
    with pytest.raises(ValueError):
        DeviceSampler(_data(1, [63, 80]), _data(2, [80]), n_frames=64)             # an utterance shorter than n_frames
    with pytest.raises(ValueError):
        DeviceSampler(_data(1, [80]), _data(2, [80]), n_frames=64, max_mask_len=65)
    sm = DeviceSampler(_data(1, [80]), _data(2, [80]), n_frames=64)
    with pytest.raises(RuntimeError):
        sm.draw_into(*[torch.empty(2, 80, 64) for _ in range(4)])                 # host tensors


def test_engine_step_sampled_equals_step_on_the_same_minibatch():
    """step_sampled() lets the kernel write into the engine's static inputs; the iteration must be the one step() runs on the
    same four tensors (bit-exact in deterministic mode)."""
    import mcvc_oracle as orc
    from mask_cyclegan_vc import _hip
    from mask_cyclegan_vc.engine import D_NAMES, G_NAMES, TrainEngine
    from mask_cyclegan_vc.model import Discriminator, Generator
    from mask_cyclegan_vc.schedule import StepSchedule
    L = _hip.lib()
    was = L.mcvc_set_deterministic(1)
    try:
        dA, dB = _data(5, [90, 64, 130]), _data(6, [100, 72])
        finals = []
        for mode in ("sampled", "explicit"):
            nets = {}
            for i, n in enumerate(orc.NET_ORDER):
                mdl = Generator() if i < 2 else Discriminator()
                mdl.load_state_dict(orc.filler_params("G" if i < 2 else "D", 610 + i), strict=True)
                nets[n] = mdl.cuda()
            eng = TrainEngine(nets, 2, 64, schedule=StepSchedule(batch_size=2, n_samples=4))
            sm = DeviceSampler(dA, dB, n_frames=64, max_mask_len=25, seed=77)
            losses = []
            for it in range(2):
                if mode == "sampled":
                    eng.step_sampled(sm)
                else:
                    ref, _ = so.draw_batch(dA, dB, 2, 64, 25, seed=77, step=it)
                    eng.step(*[torch.from_numpy(a).cuda() for a in ref])
                losses.append(tuple(sorted(eng.losses().items())))
            eng.flush()
            finals.append((losses, [p.detach().clone() for n in G_NAMES + D_NAMES for p in nets[n].parameters()]))
        assert finals[0][0] == finals[1][0]
        assert all(torch.equal(a, b) for a, b in zip(finals[0][1], finals[1][1]))
    finally:
        L.mcvc_set_deterministic(was)
