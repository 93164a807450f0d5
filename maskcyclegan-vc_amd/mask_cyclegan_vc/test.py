"""``python -m mask_cyclegan_vc.test`` -- drop-in for the reference inference driver (mask_cyclegan_vc/test.py:18-126).

Loads one generator checkpoint and converts every utterance of the source speaker with an all-ones mask
(test.py:92, 107).  The MelGAN vocoder decode + wav writing of the reference need network access and audio packages
(out of scope); the converted, de-normalised mel-spectrograms are written as .npy next to where the wavs would go.

Batching (new): InstanceNorm statistics run over an utterance's whole time axis, so zero-padding utterances to a common
length would change every output.  Utterances are therefore bucketed by EXACT length: a bucket of k equal-length utterances
is one batched forward (up to ``--max_batch``; every op is per-sample, so results equal the bs=1 results), buckets are
visited longest first and alternate between two HIP streams so that short utterances overlap on the chip.  ``--dtype bf16``
selects the bf16-MFMA forward (BASELINE configs[4])."""
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
        outs = [None] * len(src)
        buckets = {}
        for i, mel in enumerate(src):
            buckets.setdefault(int(np.asarray(mel).shape[1]), []).append(i)
        self.generator.prepare_inference(self.args.dtype)      # weight packs are built once, on the current stream, before the lanes fork
        streams = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        pending = []
        k = 0

        def drain(keep):
            """Write out the oldest groups until at most ``keep`` are in flight: device memory stays bounded by a few groups whatever the
            dataset size (the reference holds one utterance at a time, test.py:85-119)."""
            while len(pending) > keep:
                grp, fake, _real, ev = pending.pop(0)
                ev.synchronize()                              # the group's own stream is done with `fake` and `_real`
                host = fake.cpu().numpy()
                for j, i in enumerate(grp):
                    path = os.path.join(self.converted_dir, "%d-converted_%s.npy" % (i, tag))
                    np.save(path, denormalize_mel(host[j], mean, std).astype(np.float32))
                    outs[i] = path
        with torch.no_grad():
            for T in sorted(buckets, reverse=True):
                ids = buckets[T]
                for lo in range(0, len(ids), max(1, self.args.max_batch)):
                    grp = ids[lo:lo + max(1, self.args.max_batch)]
                    st = streams[k % 2]; k += 1
                    st.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(st):
                        real = torch.from_numpy(np.stack([np.asarray(src[i], dtype=np.float32) for i in grp])).to(self.device, non_blocking=True)
                        fake = self.generator.infer(real, None, dtype=self.args.dtype).float()      # all-ones mask (test.py:92)
                        ev = torch.cuda.Event()
                        ev.record(st)
                    pending.append((grp, fake, real, ev))     # keep `real` alive until the stream is done with it
                    drain(4)                                  # two groups per stream in flight
            drain(0)
            for st in streams:
                torch.cuda.current_stream(self.device).wait_stream(st)
        print("wrote %d converted mel-spectrograms to %s" % (len(outs), self.converted_dir))
        return outs


def main(argv=None):
    args = CycleGANTestArgParser().parse_args(argv)
    MaskCycleGANVCTesting(args).test()


if __name__ == "__main__":
    main()
