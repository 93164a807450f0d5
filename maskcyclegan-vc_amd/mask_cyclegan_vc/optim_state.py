"""torch.optim.Adam-compatible optimizer state of the training engine's flat Adam buffers (the reference's checkpoint layout:
saver/model_saver.py:46-123 stores ``optimizer.state_dict()`` of the two torch.optim.Adam instances of train.py:113-122)."""
from __future__ import annotations

import torch

G_NAMES = ("generator_A2B", "generator_B2A")
D_NAMES = ("discriminator_A", "discriminator_B", "discriminator_A2", "discriminator_B2")
_DEAD = range(14, 18)          # discriminator downSample4.* slots in named_parameters() order


def _align4(n):
    return (n + 3) & ~3


def optimizer_state_dict(eng, which):
    """``torch.optim.Adam.state_dict()``-shaped dict: per-parameter ``step/exp_avg/exp_avg_sq`` keyed by the
    position in the concatenated parameter list (G: 0..219; D: 0..79 with the dead 14-17,34-37,... absent)."""
    eng.flush()
    grp = eng.g_group if which == "G" else eng.d_group
    names = G_NAMES if which == "G" else D_NAMES
    lr = eng.sched.g_opt_lr if which == "G" else eng.sched.d_opt_lr
    state, idx, off = {}, 0, 0
    for n in names:
        ps = list(eng.nets[n].parameters())
        for i, p in enumerate(ps):
            if which == "D" and i in _DEAD:
                idx += 1
                continue
            k = p.numel()
            if grp.step > 0:
                state[idx] = {"step": torch.tensor(float(grp.step)),
                              "exp_avg": grp.exp_avg[off:off + k].view(p.shape).detach().cpu().clone(),
                              "exp_avg_sq": grp.exp_avg_sq[off:off + k].view(p.shape).detach().cpu().clone()}
            off += _align4(k)
            idx += 1
    group = {"lr": lr, "betas": tuple(eng.betas), "eps": eng.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
             "params": list(range(idx))}
    return {"state": state, "param_groups": [group]}

def load_optimizer_state_dict(eng, which, sd):
    eng.flush()
    grp = eng.g_group if which == "G" else eng.d_group
    names = G_NAMES if which == "G" else D_NAMES
    idx, off, step = 0, 0, 0
    for n in names:
        for i, p in enumerate(eng.nets[n].parameters()):
            if which == "D" and i in _DEAD:
                idx += 1
                continue
            k = p.numel()
            st = sd["state"].get(idx)
            if st is not None:
                grp.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                grp.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(float(st["step"])))
            off += _align4(k)
            idx += 1
    grp.step = step
    lr = sd["param_groups"][0]["lr"]
    if which == "G":
        eng.sched.g_opt_lr = lr
    else:
        eng.sched.d_opt_lr = lr
