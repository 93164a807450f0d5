"""MaskCycleGAN-VC inference flags (reference args/cycleGAN_test_arg_parser.py:16-26)."""
from .base_arg_parser import BaseArgParser


class CycleGANTestArgParser(BaseArgParser):
    isTrain = False
    FLAGS = [
        ("--sample_rate", dict(type=int, default=22050, help="Sampling rate of mel-spectrograms.")),
        ("--speaker_A_id", dict(type=str, default="VCC2SF3", help="Source speaker id (From VOC dataset).")),
        ("--speaker_B_id", dict(type=str, default="VCC2TF1", help="Source speaker id (From VOC dataset).")),
        ("--preprocessed_data_dir", dict(type=str, default="vcc2018_training_preprocessed/", help="Directory containing preprocessed dataset files.")),
        ("--ckpt_dir", dict(type=str, default=None, help="Path to model ckpt.")),
        ("--model_name", dict(type=str, choices=("generator_A2B", "generator_B2A"), default="generator_A2B", help="Name of model to load.")),
        # (new) MI355X inference knobs -- additive, defaults reproduce the reference's fp32 one-utterance-at-a-time results
        ("--dtype", dict(type=str, choices=("f32", "bf16"), default="f32", help="(new) arithmetic of the generator forward: f32 (reference numerics) or bf16 MFMA.")),
        ("--max_batch", dict(type=int, default=16, help="(new) utterances of identical length are converted in one batched forward of up to this many.")),
    ]
