from .model_saver import ModelSaver

__all__ = ["ModelSaver"]
