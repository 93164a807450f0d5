"""Vocoder / figure helpers of the reference (mask_cyclegan_vc/utils.py:25-65) are OUT OF SCOPE for this build:
they need the MelGAN hub model (network) plus librosa / torchaudio / cv2, none of which are on the training step path
(SURVEY.md section 2 row 8).  The names exist so reference-style imports resolve; calling them explains what to do."""


def _out_of_scope(name):
    raise NotImplementedError(
        "%s needs the MelGAN vocoder (torch.hub descriptinc/melgan-neurips) and audio packages that are not part of the "
        "MI355X hot-path build; the converted mel-spectrograms are written as .npy instead (see test.py)" % name)


def decode_melspectrogram(vocoder, melspectrogram, mel_mean, mel_std):
    _out_of_scope("decode_melspectrogram")


def get_mel_spectrogram_fig(spec, title="Mel-Spectrogram"):
    _out_of_scope("get_mel_spectrogram_fig")


def denormalize_mel(mel, mel_mean, mel_std):
    """The numeric half of the reference's decode (utils.py:36): undo the per-bin standardisation."""
    return mel * mel_std + mel_mean
