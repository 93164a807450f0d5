"""``python -m mask_cyclegan_vc.test`` -- drop-in for the reference inference driver (mask_cyclegan_vc/test.py:18-126).

Loads one generator checkpoint and converts every utterance of the source speaker with an all-ones mask
(test.py:92, 107).  The MelGAN vocoder decode + wav writing of the reference need network access and audio packages
(out of scope); the converted, de-normalised mel-spectrograms are written as .npy next to where the wavs would go."""
import os

import numpy as np
import torch

from args.cycleGAN_test_arg_parser import CycleGANTestArgParser
from saver.model_saver import ModelSaver

from .model import Generator
from .train import load_speaker
from .utils import denormalize_mel


class MaskCycleGANVCTesting(object):
    def __init__(self, args):
        self.args = args
        if not torch.cuda.is_available():
            raise RuntimeError("mask_cyclegan_vc.test (MI355X build) needs a HIP device; there is no CPU path")
        self.device = torch.device("cuda")
        self.model_name = args.model_name
        self.converted_dir = os.path.join(args.save_dir, args.name, "converted_mel")
        os.makedirs(self.converted_dir, exist_ok=True)
        self.dataset_A, self.dataset_A_mean, self.dataset_A_std = load_speaker(args.preprocessed_data_dir, args.speaker_A_id)
        self.dataset_B, self.dataset_B_mean, self.dataset_B_std = load_speaker(args.preprocessed_data_dir, args.speaker_B_id)
        self.generator = Generator().to(self.device)
        self.generator.eval()
        self.saver = ModelSaver(args)
        self.saver.load_model(self.generator, self.model_name)

    def test(self):
        a2b = self.model_name == "generator_A2B"
        src = self.dataset_A if a2b else self.dataset_B
        mean, std = (self.dataset_B_mean, self.dataset_B_std) if a2b else (self.dataset_A_mean, self.dataset_A_std)
        tag = ("%s_to_%s" % (self.args.speaker_A_id, self.args.speaker_B_id)) if a2b else ("%s_to_%s" % (self.args.speaker_B_id, self.args.speaker_A_id))
        outs = []
        with torch.no_grad():
            for i, mel in enumerate(src):
                real = torch.from_numpy(np.asarray(mel, dtype=np.float32)).unsqueeze(0).to(self.device)
                fake = self.generator(real, torch.ones_like(real))
                conv = denormalize_mel(fake[0].cpu().numpy(), mean, std)
                path = os.path.join(self.converted_dir, "%d-converted_%s.npy" % (i, tag))
                np.save(path, conv.astype(np.float32))
                outs.append(path)
        print("wrote %d converted mel-spectrograms to %s" % (len(outs), self.converted_dir))
        return outs


def main(argv=None):
    args = CycleGANTestArgParser().parse_args(argv)
    MaskCycleGANVCTesting(args).test()


if __name__ == "__main__":
    main()
