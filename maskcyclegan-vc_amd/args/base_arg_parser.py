"""Command-line surface shared by training and testing -- drop-in for the reference's
``args/base_arg_parser.py`` (flag names, defaults and ``parse_args`` side effects: seeding, result
directories, ``{train,test}_args.json``, device pick, resume-epoch resolution; reference :32-145).

Written as a declarative flag table; subclasses extend ``FLAGS`` and ``DEFAULT_OVERRIDES``.
New, additive flags of this implementation are marked (new)."""
import argparse
import json
import os
import random

import numpy as np
import torch


class BaseArgParser(object):
    isTrain = None
    FLAGS = [
        ("--name", dict(type=str, default="debug", help="Experiment name prefix.")),
        ("--batch_size", dict(type=int, default=20, help="Batch size (per GPU).")),
        ("--save_dir", dict(type=str, default="/home/results/", help="Directory for results including ckpts.")),
        ("--seed", dict(type=int, default=0, help="Random Seed.")),
        ("--gpu_ids", dict(type=str, default="0", help="Comma-separated list of GPU IDs.")),
        ("--steps_per_print", dict(type=int, default=1000, help="Samples between loss log lines.")),
        ("--epochs_per_save", dict(type=int, default=1, help="Epochs between checkpoints.")),
        ("--start_epoch", dict(type=int, default=1, help="Epoch to start training")),
        ("--load_epoch", dict(type=int, default=0, help="Default uses latest cached model if continue train or eval set")),
    ]
    DEFAULT_OVERRIDES = {}

    def __init__(self):
        self.parser = argparse.ArgumentParser(description=type(self).__name__)
        seen = set()
        for klass in reversed(type(self).__mro__):
            for flag, kw in klass.__dict__.get("FLAGS", []):
                if flag not in seen:
                    self.parser.add_argument(flag, **kw)
                    seen.add(flag)
        for klass in reversed(type(self).__mro__):
            over = klass.__dict__.get("DEFAULT_OVERRIDES", {})
            if over:
                self.parser.set_defaults(**over)

    # -- reference base_arg_parser.py:58-123
    def parse_args(self, argv=None):
        args = self.parser.parse_args(argv)
        os.environ["PYTHONHASHSEED"] = str(args.seed)
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(args.seed)
        np.random.seed(args.seed)
        if self.isTrain is not None:
            args.isTrain = self.isTrain
        run_dir = os.path.join(args.save_dir, args.name)
        os.makedirs(run_dir, exist_ok=True)
        if int(os.environ.get("RANK", "0")) == 0:                # under torchrun only rank 0 writes the run's files
            with open(os.path.join(run_dir, ("train" if args.isTrain else "test") + "_args.json"), "w") as fh:
                json.dump(vars(args), fh, indent=4, sort_keys=True)
        if args.isTrain:
            args.ckpt_dir = os.path.join(run_dir, "ckpts")
            os.makedirs(args.ckpt_dir, exist_ok=True)
        if int(os.environ.get("WORLD_SIZE", "1")) == 1:          # under torchrun every rank picks its GPU by LOCAL_RANK
            os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu_ids
        ids = [int(t) for t in args.gpu_ids.split(",") if t != ""]
        if len(ids) > 0 and torch.cuda.is_available():
            args.gpu_ids = ["cuda:%d" % i for i in range(len(ids))]
            args.device = "cuda"
        else:
            args.gpu_ids = ids
            args.device = "cpu"
        # resume-epoch resolution (reference :112-119)
        if not args.isTrain or getattr(args, "continue_train", False):
            if args.load_epoch > 0:
                args.start_epoch = args.load_epoch + 1
            elif args.start_epoch > 1:
                args.load_epoch = args.start_epoch - 1
            else:
                args.load_epoch = self.get_last_saved_epoch(args)
                args.start_epoch = args.load_epoch + 1
        return args

    @staticmethod
    def get_last_saved_epoch(args):
        """Epoch of the newest ``NNNNN_<model>.pth.tar`` in ``args.ckpt_dir`` (0 if none)."""
        names = sorted(n for n in os.listdir(args.ckpt_dir) if n.split(".", 1)[-1] == "pth.tar")
        return int(names[-1][:5]) if names else 0
