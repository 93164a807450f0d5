"""``python -m mask_cyclegan_vc.train`` -- drop-in for the reference trainer (mask_cyclegan_vc/train.py).

Same CLI (args/), same dataset files, same checkpoint layout, same per-iteration semantics (train.py:175-315); the
arithmetic runs in the HIP engine (engine.py).  Launch one process per GPU with torchrun for data-parallel training
(RCCL gradient all-reduce, parallel.py); a single process behaves exactly like the reference's single-device loop.

Input pipeline: by default minibatches are drawn ON THE DEVICE (dataset/device_sampler.py: HBM-resident utterance bank, one
kernel launch per iteration, no DataLoader and no H2D copy); ``--host_sampler`` keeps the reference's RNG-exact
VCDataset + DataLoader path (dataset/vc_dataset.py).  Validation (reference train.py:317-358): every ``--epochs_per_plot``
epochs the last training pair and the full-utterance conversions of the first validation pair are written as ``.npy``
spectrograms under ``<save_dir>/<name>/validation/`` -- the reference renders them through librosa figures and the MelGAN
vocoder into TensorBoard, which need network access and audio packages (out of scope)."""
import os
import pickle

import numpy as np
import torch
import torch.utils.data as data

from args.cycleGAN_train_arg_parser import CycleGANTrainArgParser
from dataset.device_sampler import DeviceSampler
from dataset.vc_dataset import VCDataset
from logger.train_logger import TrainLogger
from saver.model_saver import ModelSaver

from .engine import D_NAMES, G_NAMES, TrainEngine
from .model import Discriminator, Generator
from .parallel import FlatGradReducer, init_from_env
from .schedule import StepSchedule
from .utils import denormalize_mel

NET_NAMES = G_NAMES + D_NAMES          # construction order of the reference (train.py:103-110)
LOSS_LAG = 2                           # the per-iteration loss readback (reference train.py:302-304) returns the iteration issued two step()s ago


def load_speaker(preprocessed_dir, speaker_id):
    """<dir>/<spk>/<spk>_normalized.pickle (list of float32 [80,T]) and <spk>_norm_stat.npz (reference train.py:51-64)."""
    base = os.path.join(preprocessed_dir, speaker_id)
    with open(os.path.join(base, "%s_normalized.pickle" % speaker_id), "rb") as fh:
        mels = pickle.load(fh)
    stat = np.load(os.path.join(base, "%s_norm_stat.npz" % speaker_id))
    return mels, stat["mean"], stat["std"]


class MaskCycleGANVCTraining(object):
    def __init__(self, args):
        self.args = args
        self.rank, self.world, self.local_rank = init_from_env()
        # control-plane collectives (the collective abort of train()'s fault probe) run over gloo on host tensors: they must not queue
        # behind -- or synchronise with -- the device work of the step in flight
        self._ctl_group = None
        if self.world > 1:
            import torch.distributed as dist
            self._ctl_group = dist.new_group(backend="gloo")
        if not torch.cuda.is_available():
            raise RuntimeError("mask_cyclegan_vc.train (MI355X build) needs a HIP device; there is no CPU path")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.num_epochs = args.num_epochs
        self.start_epoch = args.start_epoch
        self.mini_batch_size = args.batch_size
        self.epochs_per_save = args.epochs_per_save
        self.epochs_per_plot = args.epochs_per_plot
        self.dataset_A, self.dataset_A_mean, self.dataset_A_std = load_speaker(args.preprocessed_data_dir, args.speaker_A_id)
        self.dataset_B, self.dataset_B_mean, self.dataset_B_std = load_speaker(args.preprocessed_data_dir, args.speaker_B_id)
        self.n_samples = len(self.dataset_A)
        if self.rank == 0:
            print("n_samples = %d" % self.n_samples)
        if self.world > 1:                       # every rank draws its own minibatches
            np.random.seed(args.seed + self.rank)
        self.dataset = VCDataset(datasetA=self.dataset_A, datasetB=self.dataset_B, n_frames=args.num_frames, max_mask_len=args.max_mask_len)
        self.train_dataloader = None
        self.sampler = None
        if args.host_sampler:                    # the reference's path, bit-identical draws for a given --seed (vc_dataset.py:19-77)
            self.train_dataloader = data.DataLoader(dataset=self.dataset, batch_size=self.mini_batch_size, shuffle=True, drop_last=False)
        else:
            self.sampler = DeviceSampler(self.dataset_A, self.dataset_B, n_frames=args.num_frames, max_mask_len=args.max_mask_len,
                                         device=self.device, seed=args.seed + 7919 * self.rank)
            # the sampler's random stream is keyed by (seed, minibatch number): a resumed run continues with the minibatch the
            # uninterrupted run would have drawn next (epochs have ceil(len(dataset) / batch_size) iterations), so
            # `--continue_train --start_epoch k` reproduces epochs k.. of a straight run exactly (tests/test_hip_cli.py)
            iters_per_epoch = -(-len(self.dataset) // self.mini_batch_size)
            self.sampler.step = (args.start_epoch - 1) * iters_per_epoch
        # validation pair (reference train.py:86-96: VCDataset(valid=True), batch_size 1, no shuffle -> the first utterances, whole)
        self.validation_dataset = VCDataset(datasetA=self.dataset_A, datasetB=self.dataset_B, n_frames=args.num_frames_validation,
                                            max_mask_len=args.max_mask_len, valid=True)
        self.logger = TrainLogger(args, len(self.dataset), world_size=self.world, rank=self.rank)
        self.saver = ModelSaver(args)            # like the reference, max_ckpts is not passed: nothing is pruned
        # six networks in the reference's construction order; parse_args() seeded torch, so default init matches
        torch.manual_seed(args.seed)
        nets = {}
        for n in NET_NAMES:
            nets[n] = (Generator() if n in G_NAMES else Discriminator()).to(self.device)
            setattr(self, n, nets[n])
        self.nets = nets
        sched = StepSchedule(generator_lr=args.generator_lr, discriminator_lr=args.discriminator_lr, num_epochs=args.num_epochs,
                             n_samples=self.n_samples, batch_size=self.mini_batch_size, decay_after=args.decay_after,
                             stop_identity_after=args.stop_identity_after, cycle_loss_lambda=args.cycle_loss_lambda,
                             identity_loss_lambda=args.identity_loss_lambda, start_epoch=args.start_epoch,
                             dataset_len=len(self.dataset), world_size=self.world)
        sched.global_step = self.logger.global_step
        self.engine = TrainEngine(nets, self.mini_batch_size, args.num_frames, schedule=sched,
                                  reducer=FlatGradReducer(bucket_bytes=args.allreduce_bucket_mb << 20))
        self.generator_optimizer = self.engine.optimizer("G")
        self.discriminator_optimizer = self.engine.optimizer("D")
        if args.continue_train:                  # reference train.py:125-137
            self.saver.load_model(self.generator_A2B, "generator_A2B", None, self.generator_optimizer)
            self.saver.load_model(self.generator_B2A, "generator_B2A", None, None)
            self.saver.load_model(self.discriminator_A, "discriminator_A", None, self.discriminator_optimizer)
            self.saver.load_model(self.discriminator_B, "discriminator_B", None, None)
            self.saver.load_model(self.discriminator_A2, "discriminator_A2", None, None)
            self.saver.load_model(self.discriminator_B2, "discriminator_B2", None, None)
            self.engine.repack(NET_NAMES)

    def save_all(self, epoch):
        """Six files per epoch; the G optimizer state rides in both generator files and the D state in all four
        discriminator files (reference train.py:361-373)."""
        if self.rank != 0:
            return
        for n in NET_NAMES:
            opt = self.generator_optimizer if n in G_NAMES else self.discriminator_optimizer
            self.saver.save(epoch, self.nets[n], opt, None, self.device, n)

    def _epoch_batches(self):
        """Yield one iteration's work at a time: a callable that runs the step.  Epoch length and the short last batch follow
        the reference's DataLoader(shuffle=True, drop_last=False) over len(dataset) samples."""
        if self.train_dataloader is not None:
            for real_A, mask_A, real_B, mask_B in self.train_dataloader:
                batch = [t.to(self.device, dtype=torch.float).contiguous() for t in (real_A, mask_A, real_B, mask_B)]
                yield lambda b=batch: self.engine.step(*b)
        else:
            left = len(self.dataset)
            while left > 0:
                nb = min(self.mini_batch_size, left)
                left -= nb
                yield lambda nb=nb: self.engine.step_sampled(self.sampler, nb)

    def validate(self):
        """Reference train.py:317-358 without the figure / vocoder back-ends: the last training pair's first sample
        (real_A, generated_A, real_B, generated_B of the discriminator phase) and the full-utterance conversions of the first
        validation pair (G_A2B(real_full_A, ones), G_B2A(real_full_B, ones)), de-normalised, as .npy spectrograms."""
        eng = self.engine
        eng.flush()
        B = eng.B
        arrays = {
            "real_A_spec": eng.static_in[0][0].cpu().numpy(), "real_B_spec": eng.static_in[2][0].cpu().numpy(),
            "fake_A_spec": eng.d_in["discriminator_A"][B].cpu().numpy(),      # generated_A = G_B2A(real_B, mask_B), train.py:259
            "fake_B_spec": eng.d_in["discriminator_B"][B].cpu().numpy(),      # generated_B = G_A2B(real_A, mask_A), train.py:267
        }
        full_A, full_B = self.validation_dataset[0]
        with torch.no_grad():
            xa = torch.from_numpy(np.asarray(full_A, dtype=np.float32)).unsqueeze(0).to(self.device)
            xb = torch.from_numpy(np.asarray(full_B, dtype=np.float32)).unsqueeze(0).to(self.device)
            fake_full_B = self.generator_A2B(xa, torch.ones_like(xa))[0].cpu().numpy()
            fake_full_A = self.generator_B2A(xb, torch.ones_like(xb))[0].cpu().numpy()
        arrays["real_speaker_A_mel"] = denormalize_mel(np.asarray(full_A), self.dataset_A_mean, self.dataset_A_std)
        arrays["fake_speaker_A_mel"] = denormalize_mel(fake_full_A, self.dataset_A_mean, self.dataset_A_std)
        arrays["real_speaker_B_mel"] = denormalize_mel(np.asarray(full_B), self.dataset_B_mean, self.dataset_B_std)
        arrays["fake_speaker_B_mel"] = denormalize_mel(fake_full_B, self.dataset_B_mean, self.dataset_B_std)
        return self.logger.log_spectrograms(os.path.join(self.args.save_dir, self.args.name, "validation"), arrays)

    def train(self):
        """The reference's loop (train.py:175-315).  The engine's pipelined step completes an iteration's discriminator phase beside the
        NEXT iteration's generator phase, and the host reads a COMPLETE iteration LOSS_LAG = 2 ``step()``s behind so that it never waits for
        work it has just queued: the logger is fed iteration t's losses (both of them, like the reference's ``.item()`` reads at :303) right
        after iteration t+2 has been issued, the epoch's last iterations after the epoch's flush -- the log lines are the reference's, line for
        line, two iterations late (the NaN probe below lags by the same two iterations; ``bench.py --sync-losses`` prices the reference-exact
        readback at +15 %: INTEGRATION.md)."""
        done = 0
        for epoch in range(self.start_epoch, self.num_epochs + 1):
            self.logger.start_epoch()
            owed = 0                                                       # iterations issued whose losses are not logged yet

            def log_one(lo):
                # A persistent trunk launch that gave up waiting poisons its pass with NaN (csrc/trunk.h): that reaches every loss term of
                # the iteration, so the (already host-resident) losses are the per-iteration fault probe.  The losses are LOSS_LAG step()s
                # behind (iteration t is checked after t + 2 has been issued), so the poisoned update may already be in the weights: what the
                # abort guarantees is that no CHECKPOINT is written from them.  Data parallel: the losses are rank-local -- only the
                # poisoned rank sees the NaN at once, the others one iteration later through the all-reduced gradients -- so the verdict is
                # taken collectively (one integer over the gloo control group: no device work, no stream sync) and every rank leaves the
                # loop in the same iteration instead of blocking in the next collective until a watchdog kills it.
                bad = not (np.isfinite(lo["g_loss"]) and np.isfinite(lo["d_loss"]))
                if self._ctl_group is not None:
                    import torch.distributed as dist
                    flag = torch.tensor([1 if bad else 0], dtype=torch.int32)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._ctl_group)
                    bad_any = bool(flag.item())
                else:
                    bad_any = bad
                if bad_any:
                    self.engine.check_faults(raise_on_fault=bad)          # (names the cause on the rank that faulted)
                    raise FloatingPointError("non-finite losses on %s (g_loss=%r, d_loss=%r here): aborting without saving; resume from the "
                                             "last checkpoint with --continue_train" % ("this rank" if bad else "another rank", lo["g_loss"], lo["d_loss"]))
                self.logger.log_iter(loss_dict={"g_loss": lo["g_loss"], "d_loss": lo["d_loss"]})
                self.logger.end_iter()
            for run_step in self._epoch_batches():
                self.logger.start_iter()
                try:
                    run_step()                                             # G phase, (previous) D phase, lr / lambda bookkeeping
                except Exception:
                    # any other failure on this rank (data error, out of memory): publish it through the same control collective the other
                    # ranks enter in log_one, so they abort with this rank instead of blocking in gloo until its timeout
                    if self._ctl_group is not None:
                        import torch.distributed as dist
                        dist.all_reduce(torch.tensor([1], dtype=torch.int32), op=dist.ReduceOp.MAX, group=self._ctl_group)
                    raise
                owed += 1
                # Host read of a COMPLETE iteration: the LOSS_LAG-th newest.  Reading the newest (lag 1) waits for the discriminator phase this
                # step() has just queued -- the host then issues the next iteration into a drained GPU (0.3 - 0.5 ms of a 6 ms step at bs=1,
                # DESIGN section 5); with lag 2 the copy being waited for was published one step() earlier and the host stays ahead.
                lo = self.engine.losses(lagged=LOSS_LAG)
                in_flight = LOSS_LAG if self.engine._pending_D is not None else 0   # pipelined: iterations issued whose losses are not readable yet
                if lo is not None and owed > in_flight:
                    log_one(lo)
                    owed -= 1
                done += 1
                if self.args.max_iters and done >= self.args.max_iters:
                    break
            while owed > 1:                                                # the epoch's tail: iterations published but not read yet, oldest first
                lo = self.engine.losses(lagged=owed - 1) if self.engine._pending_D is not None else None
                if lo is None:                                             # (nothing pending / read before: only the newest is left to report)
                    break
                log_one(lo)
                owed -= 1
            if owed:
                log_one(self.engine.losses())                              # completes the epoch's last iteration (flush) and reads it
                owed -= 1
            if done and self.epochs_per_plot and epoch % self.epochs_per_plot == 0 and self.rank == 0:
                self.validate()                                            # train.py:317
            if epoch % self.epochs_per_save == 0:
                self.engine.flush()                                        # a deferred (data-parallel) D update must land first
                self.engine.check_faults()                                 # (device sync; never checkpoint a poisoned state)
                self.save_all(epoch)
            self.logger.end_epoch()
            if self.args.max_iters and done >= self.args.max_iters:
                break
        self.engine.flush()

    def close(self):
        """Tear down the process group (RCCL warns / may hang at interpreter exit otherwise)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            torch.cuda.synchronize(self.device)
            dist.barrier()
            dist.destroy_process_group()


def main(argv=None):
    args = CycleGANTrainArgParser().parse_args(argv)
    job = MaskCycleGANVCTraining(args)
    try:
        job.train()
    finally:
        job.close()


if __name__ == "__main__":
    main()
