"""MaskCycleGAN-VC training flags (reference args/cycleGAN_train_arg_parser.py:18-51)."""
from .train_arg_parser import TrainArgParser


class CycleGANTrainArgParser(TrainArgParser):
    FLAGS = [
        ("--sample_rate", dict(type=int, default=22050, help="Sampling rate of mel-spectrograms.")),
        ("--speaker_A_id", dict(type=str, default="28", help="Source speaker id (From VOC dataset).")),
        ("--speaker_B_id", dict(type=str, default="DCB_se2_ag3_m_02_1", help="Target speaker id (From CORAAL dataset).")),
        ("--preprocessed_data_dir", dict(type=str, default="vcc2018_training_preprocessed/", help="Directory containing preprocessed dataset files.")),
        ("--generator_lr", dict(type=float, default=2e-4, help="Initial generator learning rate.")),
        ("--discriminator_lr", dict(type=float, default=1e-4, help="Initial discrminator learning rate.")),
        ("--cycle_loss_lambda", dict(type=float, default=10, help="Lambda value for cycle consistency loss.")),
        ("--identity_loss_lambda", dict(type=float, default=5, help="Lambda value for identity loss.")),
        ("--epochs_per_plot", dict(type=int, default=2, help="Epochs per save plot.")),
        ("--num_frames", dict(type=int, default=64, help="Num frames per training sample.")),
        ("--num_frames_validation", dict(type=int, default=320, help="Num frames per validation sample.")),
        ("--max_mask_len", dict(type=int, default=32, help="Maximum length of mask for Mask-CycleGAN-VC.")),
        # (new) MI355X / data-parallel knobs -- additive.  NOTE the default input path is the on-device sampler: the reference's
        # distributions from a counter-based random stream, NOT the reference's numpy draws -- pass --host_sampler for minibatches that
        # are bit-identical to the reference's for a given --seed (dataset/vc_dataset.py)
        ("--allreduce_bucket_mb", dict(type=int, default=64, help="(new) RCCL gradient all-reduce bucket size in MiB.")),
        ("--max_iters", dict(type=int, default=0, help="(new) stop after this many iterations (0 = run all epochs).")),
        ("--host_sampler", dict(action="store_true", help="(new) draw minibatches on the host with the reference's RNG-exact VCDataset + DataLoader "
                                                          "instead of the on-device sampler (same distributions, no H2D copies).")),
    ]
    DEFAULT_OVERRIDES = dict(batch_size=1, num_epochs=50, decay_after=1e4, start_epoch=1, steps_per_print=100, num_frames=64)
