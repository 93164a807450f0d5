"""Training flags (reference args/train_arg_parser.py:15-26)."""
from .base_arg_parser import BaseArgParser


class TrainArgParser(BaseArgParser):
    isTrain = True
    FLAGS = [
        ("--num_epochs", dict(type=int, default=6500, help="Number of epochs to train.")),
        ("--decay_after", dict(type=float, default=2e5, help="Decay learning rate after n iterations.")),
        ("--stop_identity_after", dict(type=float, default=1e4, help="Stop using identity loss after n iterations.")),
        ("--max_ckpts", dict(type=int, default=3, help="Max ckpts to save.")),
        ("--continue_train", dict(action="store_true", help="continue training: load the latest model")),
    ]
