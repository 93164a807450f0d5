#!/usr/bin/env python3
"""Pin the STRUCTURE of checkpoint files written by the reference's own ``saver.ModelSaver.save`` (SURVEY.md section 8c
item 8: the files are 25-98 MB, so key names / dtypes / shapes / param_groups are recorded as JSON instead of the bytes).

Runs only in the build container (reference mounted read-only at /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt.py

One torch.optim.Adam step with synthetic gradients gives the optimizer per-parameter state exactly where the reference's
training gives it (discriminator ``downSample4`` parameters never receive a gradient -> no state); the
unmodified ``ModelSaver.save`` then writes generator_A2B / discriminator_A files which are loaded back and described.
"""
import json
import os
import sys
import tempfile
from argparse import Namespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


def describe(v):
    if torch.is_tensor(v):
        return {"tensor": str(v.dtype).replace("torch.", ""), "shape": list(v.shape), "device": v.device.type}
    if isinstance(v, (list, tuple)):
        return {"type": type(v).__name__, "len": len(v), "elem": type(v[0]).__name__ if len(v) else None}
    return {"type": type(v).__name__, "value": v if isinstance(v, (int, float, bool, str, type(None))) else None}


def main():
    sys.path.insert(0, REF)
    from mask_cyclegan_vc.model import Generator, Discriminator   # reference, unmodified
    from saver.model_saver import ModelSaver                       # reference, unmodified
    torch.manual_seed(0)
    gens = [Generator(), Generator()]
    discs = [Discriminator() for _ in range(4)]
    g_params = [p for g in gens for p in g.parameters()]
    d_params = [p for d in discs for p in d.parameters()]
    g_opt = torch.optim.Adam(g_params, lr=2e-4, betas=(0.5, 0.999))     # reference train.py:119-122
    d_opt = torch.optim.Adam(d_params, lr=1e-4, betas=(0.5, 0.999))
    for p in g_params:
        p.grad = torch.ones_like(p)
    for d in discs:
        for name, p in d.named_parameters():
            if not name.startswith("downSample4."):               # never used in forward -> grad stays None in real training
                p.grad = torch.ones_like(p)
    g_opt.step(); d_opt.step()
    out = {"torch": torch.__version__}
    with tempfile.TemporaryDirectory() as tmp:
        saver = ModelSaver(Namespace(ckpt_dir=tmp, load_epoch=3, gpu_ids=["cpu"]))
        for name, model, opt in (("generator_A2B", gens[0], g_opt), ("discriminator_A", discs[0], d_opt)):
            saver.save(3, model, opt, None, "cpu", name)
            fn = "00003_%s.pth.tar" % name
            assert os.path.exists(os.path.join(tmp, fn))
            ck = torch.load(os.path.join(tmp, fn), map_location="cpu", weights_only=False)
            st = ck["optimizer"]["state"]
            first = st[sorted(st)[0]]
            out[name] = {
                "file_name": fn,
                "top_level_keys": list(ck.keys()),
                "ckpt_info": ck["ckpt_info"], "model_class": ck["model_class"], "lr_scheduler": ck["lr_scheduler"],
                "model_state_type": type(ck["model_state"]).__name__,
                "model_state": {k: describe(v) for k, v in ck["model_state"].items()},
                "model_state_order": list(ck["model_state"].keys()),
                "optimizer_keys": list(ck["optimizer"].keys()),
                "optimizer_state_indices": sorted(st.keys()),
                "optimizer_state_entry": {k: describe(v) for k, v in first.items()},
                "optimizer_state_shapes": {str(i): list(st[i]["exp_avg"].shape) for i in sorted(st)},
                "param_groups_len": len(ck["optimizer"]["param_groups"]),
                "param_group": {k: describe(v) for k, v in ck["optimizer"]["param_groups"][0].items()},
            }
    json.dump(out, open(os.path.join(HERE, "ckpt_structure.json"), "w"), indent=0, sort_keys=True)
    print("wrote ckpt_structure.json:", {k: len(v["model_state"]) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
