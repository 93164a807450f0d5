from .base_arg_parser import BaseArgParser
from .train_arg_parser import TrainArgParser
from .cycleGAN_train_arg_parser import CycleGANTrainArgParser
from .cycleGAN_test_arg_parser import CycleGANTestArgParser

__all__ = ["BaseArgParser", "TrainArgParser", "CycleGANTrainArgParser", "CycleGANTestArgParser"]
