from .train_logger import TrainLogger

__all__ = ["TrainLogger"]
