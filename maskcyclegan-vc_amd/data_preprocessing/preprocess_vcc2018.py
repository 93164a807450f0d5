"""Writer of the preprocessed-dataset format the trainer reads (reference data_preprocessing/preprocess_vcc2018.py:26-85).

The reference turns ``.wav`` files into 80-bin mel-spectrograms with the MelGAN vocoder's front-end (``torch.hub``
descriptinc/melgan-neurips + librosa: network and audio packages, out of scope here -- SURVEY.md section 2 row 9), then
standardises per bin over the whole speaker and writes two files.  This module owns everything AFTER the wav -> mel step,
so mel-spectrograms produced by any front-end (``.npy`` files, one ``[80, T]`` array per utterance) become a dataset that
both this trainer and the reference trainer load:

    <cache>/<spk>/<spk>_normalized.pickle   list of float32 [80, T_i], (mel - mean) / std          (:40-47, :83)
    <cache>/<spk>/<spk>_norm_stat.npz       mean, std: [80, 1]; std = np.std(...) + 1e-9          (:36-38, :78-80)

Utterances shorter than 64 frames are dropped like the reference does (:33).

    python -m data_preprocessing.preprocess_vcc2018 --mel_directory mels/ --preprocessed_data_directory out/ --speaker_ids A B
"""
import argparse
import glob
import os
import pickle

import numpy as np

MIN_FRAMES = 64          # training sample = 64 randomly cropped frames (reference :33)


def normalize_mels(mel_list):
    """-> (list of standardised float32 [80,T_i], mean [80,1], std [80,1]); reference normalize_mel :35-47."""
    mel_list = [np.asarray(m) for m in mel_list if np.asarray(m).shape[-1] >= MIN_FRAMES]
    if not mel_list:
        raise ValueError("no utterance has >= %d frames" % MIN_FRAMES)
    cat = np.concatenate(mel_list, axis=1)
    mean = np.mean(cat, axis=1, keepdims=True)
    std = np.std(cat, axis=1, keepdims=True) + 1e-9
    return [((m - mean) / std).astype(np.float32) for m in mel_list], mean, std


def save_preprocessed(cache_folder, speaker_id, mel_list):
    """Standardise and write the two files of one speaker (reference preprocess_dataset :62-85)."""
    normalized, mean, std = normalize_mels(mel_list)
    d = os.path.join(cache_folder, speaker_id)
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, "%s_norm_stat.npz" % speaker_id), mean=mean, std=std)
    with open(os.path.join(d, "%s_normalized.pickle" % speaker_id), "wb") as fh:
        pickle.dump(normalized, fh)
    return d


def main(argv=None):
    ap = argparse.ArgumentParser(description="mel-spectrogram .npy files -> preprocessed speaker folders")
    ap.add_argument("--mel_directory", type=str, required=True, help="<dir>/<speaker_id>/**/*.npy, one [80,T] array per utterance")
    ap.add_argument("--preprocessed_data_directory", type=str, default="vcc2018_preprocessed/vcc2018_training")
    ap.add_argument("--speaker_ids", nargs="+", type=str, required=True)
    args = ap.parse_args(argv)
    for spk in args.speaker_ids:
        files = sorted(glob.glob(os.path.join(args.mel_directory, spk, "**", "*.npy"), recursive=True))
        d = save_preprocessed(args.preprocessed_data_directory, spk, [np.load(f) for f in files])
        print("Preprocessed and saved data for speaker: %s (%d files) -> %s" % (spk, len(files), d))


if __name__ == "__main__":
    main()
