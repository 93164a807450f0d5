"""Host-side step bookkeeping of the reference trainer, restated exactly (quirks included).

Reference: ``mask_cyclegan_vc/train.py:67-74`` (decay constants), ``:139-155`` (``adjust_lr_rate``),
``:307-315`` (call site) and the counters owned by ``logger/train_logger.py:170-173`` /
``logger/base_logger.py:53-56`` (``global_step`` advances by the *batch size* per iteration and is
re-derived as ``(start_epoch-1) * len(dataset)`` on resume).

Quirk kept for drop-in parity: after ``decay_after`` the call site passes the
*generator* optimizer to both ``adjust_lr_rate`` calls, so the generator optimizer ends every
iteration holding the decayed *discriminator* learning rate and the discriminator optimizer's
learning rate never changes.
"""
from __future__ import annotations


class StepSchedule:
    def __init__(self, generator_lr=2e-4, discriminator_lr=1e-4, num_epochs=50, n_samples=1, batch_size=1,
                 decay_after=1e4, stop_identity_after=1e4, cycle_loss_lambda=10.0, identity_loss_lambda=5.0,
                 start_epoch=1, dataset_len=None, world_size=1):
        self.generator_lr = generator_lr                     # python-float attributes, train.py:35-36
        self.discriminator_lr = discriminator_lr
        self.decay_after = decay_after
        self.stop_identity_after = stop_identity_after
        self.cycle_loss_lambda = cycle_loss_lambda
        self.identity_loss_lambda = identity_loss_lambda
        self.batch_size = batch_size
        self.world_size = world_size
        denom = float(num_epochs * (n_samples // batch_size))                 # train.py:68-73
        self.generator_lr_decay = generator_lr / denom
        self.discriminator_lr_decay = discriminator_lr / denom
        # what the two optimizers actually hold in param_groups[0]['lr']
        self.g_opt_lr = generator_lr
        self.d_opt_lr = discriminator_lr
        dataset_len = n_samples if dataset_len is None else dataset_len
        # base_logger.py:53-56: round_down((epoch-1)*dataset_len, batch_size), python round().  Under data parallelism an
        # epoch advances global_step by ~dataset_len * world_size (every rank consumes batch_size samples per iteration,
        # end_iteration below), so the re-derived resume value carries the same factor.
        self.global_step = int(batch_size * round(float((start_epoch - 1) * dataset_len) / batch_size)) * world_size

    def end_iteration(self):
        """Call once per iteration after the optimizer steps (train.py:302-315)."""
        # logger.end_iter(): iter and global_step advance by batch_size (train_logger.py:170-173);
        # under data parallelism every rank consumed batch_size samples
        self.global_step += self.batch_size * self.world_size
        if self.global_step > self.decay_after:
            # adjust_lr_rate(generator_optimizer, generator=True)
            self.generator_lr = max(0.0, self.generator_lr - self.generator_lr_decay)
            self.g_opt_lr = self.generator_lr
            # adjust_lr_rate(generator_optimizer, generator=False)   <-- the reference's call-site bug
            self.discriminator_lr = max(0.0, self.discriminator_lr - self.discriminator_lr_decay)
            self.g_opt_lr = self.discriminator_lr
        if self.global_step > self.stop_identity_after:
            self.identity_loss_lambda = 0
