"""Text log + counters of the reference's TrainLogger (logger/base_logger.py:48-91, train_logger.py:121-213).

The counters matter for the hot path: ``iter`` / ``global_step`` advance by the batch size per iteration and
``global_step`` (re-derived on resume as round_down((start_epoch-1)*len(dataset), batch_size)) gates the LR decay
and the identity-loss cut-off.  TensorBoard is optional here (tensorboardX is not a dependency): scalars go to the
SummaryWriter only if one can be imported."""
import os
from datetime import datetime
from time import time


class _Meter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.sum = 0.0
        self.count = 0
        self.avg = 0.0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _summary_writer(log_dir):
    try:
        from tensorboardX import SummaryWriter          # optional
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return None


class TrainLogger(object):
    def __init__(self, args, dataset_len, world_size=1, rank=0):
        self.args = args
        self.rank = rank                         # data parallel: only rank 0 writes the .log / stdout lines / TensorBoard events
        self.batch_size = args.batch_size
        self.world_size = world_size
        self.dataset_len = dataset_len
        self.save_dir = args.save_dir
        self.steps_per_print = args.steps_per_print
        self.num_epochs = args.num_epochs
        self.summary_writer = None
        if rank == 0:
            self.summary_writer = _summary_writer(os.path.join(args.save_dir, "logs", args.name + "_" + datetime.now().strftime("%y%m%d_%H%M%S")))
        self.log_path = os.path.join(self.save_dir, args.name, "%s.log" % args.name)
        self.epoch = args.start_epoch
        self.iter = 0
        raw = (self.epoch - 1) * dataset_len
        # (x world_size: under data parallelism every iteration advances global_step by batch_size * world_size, end_iter)
        self.global_step = int(self.batch_size * round(float(raw) / self.batch_size)) * world_size
        self.iter_start_time = None
        self.epoch_start_time = None
        self.loss_meters = None

    def write(self, message, print_to_stdout=True):
        if self.rank != 0:
            return
        with open(self.log_path, "a") as fh:
            fh.write(message + "\n")
        if print_to_stdout:
            print(message)

    def _scalars(self, d):
        if self.summary_writer is not None:
            for k, v in d.items():
                self.summary_writer.add_scalar(k.replace("_", "/"), v, self.global_step)

    def log_metrics(self, metrics):
        for k, v in metrics.items():
            self.write("[%s: %s]" % (k, v))
        self._scalars(metrics)

    def log_spectrograms(self, out_dir, arrays):
        """Validation dump (reference train.py:317-358 logs figures / audio to TensorBoard through librosa + the MelGAN
        vocoder, neither available here): the same tensors as float32 ``.npy`` files, one per name, tagged with the epoch."""
        if self.rank != 0:
            return []
        import numpy as np
        os.makedirs(out_dir, exist_ok=True)
        paths = []
        for name, arr in arrays.items():
            path = os.path.join(out_dir, "epoch%05d_%s.npy" % (self.epoch, name))
            np.save(path, np.asarray(arr, dtype=np.float32))
            paths.append(path)
        self.write("[validation: wrote %d spectrograms to %s]" % (len(paths), out_dir))
        return paths

    def start_epoch(self):
        self.epoch_start_time = time()
        self.iter = 0
        self.write("[start of epoch %d]" % self.epoch)

    def start_iter(self):
        self.iter_start_time = time()

    def log_iter(self, loss_dict={}):
        if self.loss_meters is None:
            self.loss_meters = {k: _Meter() for k in loss_dict}
        for k, m in self.loss_meters.items():
            m.update(loss_dict[k], self.batch_size)
        if self.iter % self.steps_per_print == 0:
            msg = "(epoch: %d, iter: %d, time: %.3f) " % (self.epoch, self.iter, (time() - self.iter_start_time) / self.batch_size)
            msg += "".join("%s: %.3f " % (k, m.avg) for k, m in self.loss_meters.items())
            self._scalars({k: m.avg for k, m in self.loss_meters.items()})
            for m in self.loss_meters.values():
                m.reset()
            self.write(msg)

    def end_iter(self):
        self.iter += self.batch_size
        self.global_step += self.batch_size * self.world_size

    def end_epoch(self, metrics=None):
        self.write("[end of epoch %d/%d, epoch time: %.2g]" % (self.epoch, self.num_epochs, time() - self.epoch_start_time))
        if metrics:
            self.log_metrics(metrics)
        self.epoch += 1

    def is_finished_training(self):
        return 0 < self.num_epochs < self.epoch
