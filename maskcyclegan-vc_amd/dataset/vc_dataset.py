"""Paired mel-spectrogram dataset with filling-in-frames masks -- drop-in for the reference's
``dataset/vc_dataset.py`` (same constructor, same outputs, same consumption of the *global* numpy RNG, so a
seeded run draws bit-identical crops and masks; pinned by tests/golden/dataset_draws.npz).

Reference behaviour that is kept on purpose (vc_dataset.py:19-77): every ``__getitem__`` re-shuffles all
utterance indices and crops / masks EVERY pair before returning element ``index``; ``mask_B`` takes its shape
from the A crop.  The training CLI uses this RNG-exact host path only behind ``--host_sampler``; its default is the
on-device sampler (``dataset/device_sampler.py``: same distributions, one kernel launch per minibatch, no H2D copies)."""
import numpy as np
from torch.utils.data.dataset import Dataset


class VCDataset(Dataset):
    def __init__(self, datasetA, datasetB=None, n_frames=64, max_mask_len=25, valid=False):
        self.datasetA = datasetA
        self.datasetB = datasetB
        self.n_frames = n_frames
        self.valid = valid
        self.max_mask_len = max_mask_len

    def __len__(self):
        return len(self.datasetA) if self.datasetB is None else min(len(self.datasetA), len(self.datasetB))

    def _crop_and_mask(self, utt, like=None):
        T = self.n_frames
        total = utt.shape[1]
        assert total >= T
        lo = np.random.randint(total - T + 1)
        crop = utt[:, lo:lo + T]
        size = np.random.randint(0, self.max_mask_len)
        assert T > size
        start = np.random.randint(0, T - size)
        mask = np.ones_like(crop if like is None else like)
        mask[:, start:start + size] = 0.
        return crop, mask

    def __getitem__(self, index):
        A, B = self.datasetA, self.datasetB
        if self.valid:
            return A[index] if B is None else (A[index], B[index])
        n = min(len(A), len(B))
        self.length = n
        order_A = np.arange(len(A))
        order_B = np.arange(len(B))
        np.random.shuffle(order_A)
        np.random.shuffle(order_B)
        out = [[], [], [], []]
        for ia, ib in zip(order_A[:n], order_B[:n]):
            crop_a, mask_a = self._crop_and_mask(A[ia])
            crop_b, mask_b = self._crop_and_mask(B[ib], like=crop_a)
            for bucket, item in zip(out, (crop_a, mask_a, crop_b, mask_b)):
                bucket.append(item)
        da, ma, db, mb = (np.array(v) for v in out)
        return da[index], ma[index], db[index], mb[index]
