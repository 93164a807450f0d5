"""Checkpoint writer / reader with the reference's on-disk layout (saver/model_saver.py:46-123):

    <ckpt_dir>/{epoch:05d}_{model_name}.pth.tar = torch.save({
        'ckpt_info': {'epoch': int}, 'model_class': str, 'model_state': OrderedDict (CPU tensors, reference key set),
        'optimizer': torch.optim.Adam-shaped state_dict, 'lr_scheduler': None })

``optimizer`` may be a ``torch.optim.Optimizer`` or anything with ``state_dict()`` / ``load_state_dict()`` -- the
training engine hands in an adapter that exports its flat Adam moments in ``torch.optim.Adam`` layout, so files
written here load in the reference and vice versa."""
import os

import torch


class ModelSaver(object):
    def __init__(self, args, max_ckpts=None, metric_name=None, maximize_metric=False):
        self.args = args
        self.ckpt_dir = args.ckpt_dir
        self.max_ckpts = max_ckpts
        self.metric_name = metric_name
        self.maximize_metric = maximize_metric
        self.best_metric_val = None
        self.ckpt_names = sorted(n for n in os.listdir(self.ckpt_dir) if n.split(".", 1)[-1] == "pth.tar")

    @staticmethod
    def file_name(epoch, model_name):
        return "%s_%s.pth.tar" % (str(epoch).zfill(5), model_name)

    def save(self, epoch, model, optimizer, lr_scheduler, device, model_name):
        net = getattr(model, "module", model)                      # the reference unwraps DataParallel here
        state = {k: v.detach().to("cpu").clone() for k, v in net.state_dict().items()}
        payload = {
            "ckpt_info": {"epoch": epoch},
            "model_class": net.__class__.__name__,
            "model_state": type(net.state_dict())(state),
            "optimizer": optimizer.state_dict(),
            "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None,
        }
        path = os.path.join(self.ckpt_dir, self.file_name(epoch, model_name))
        torch.save(payload, path)
        print("Saved model to %s" % path)
        if self.max_ckpts:
            self.ckpt_names.append(path)
            if len(self.ckpt_names) > self.max_ckpts:
                oldest = os.path.join(self.ckpt_dir, self.ckpt_names.pop(0))
                os.remove(oldest)
                print("Exceeded max number of checkpoints so deleting %s" % oldest)
        return path

    def load_model(self, model, model_name=None, ckpt_path=None, optimizer=None, scheduler=None):
        if ckpt_path is None:
            if model_name and hasattr(self.args, "load_epoch"):
                ckpt_path = os.path.join(self.ckpt_dir, self.file_name(self.args.load_epoch, model_name))
            else:
                print("No checkpoint found. Failed to load load model checkpoint.")
                return None
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["model_state"])                 # strict: the reference key set (114 / 20 keys)
        if optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
        if scheduler is not None:
            scheduler.load_state_dict(ckpt["lr_scheduler"])
        print("Loaded %s from %s" % (ckpt["model_class"], ckpt_path))
        return ckpt
