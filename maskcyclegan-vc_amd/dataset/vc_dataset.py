"""Paired mel-spectrogram dataset with filling-in-frames masks -- drop-in for the reference's
``dataset/vc_dataset.py`` (same constructor, same outputs, same consumption of the *global* numpy RNG, so a
seeded run draws bit-identical crops and masks; pinned by tests/golden/dataset_draws.npz).

Reference behaviour that is kept on purpose (vc_dataset.py:19-77): every ``__getitem__`` re-shuffles all
utterance indices and crops / masks EVERY pair before returning element ``index``; ``mask_B`` takes its shape
from the A crop.  ``draw_batch`` is the (new) cheap path used when RNG-stream parity with a reference run is
not required: same distributions, O(batch) work."""
import numpy as np
from torch.utils.data.dataset import Dataset


def _fif_mask(shape, n_frames, max_mask_len, rng):
    """ones with frames [start, start+size) zeroed; size ~ U{0..max_mask_len-1}, start ~ U{0..n_frames-size-1}."""
    size = rng.randint(0, max_mask_len)
    assert n_frames > size
    start = rng.randint(0, n_frames - size)
    mask = np.ones(shape, dtype=np.float32) if shape is not None else None
    return size, start, mask


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

    # ---- (new) O(batch) sampler with the same distributions, own RandomState -------------------------------
    def draw_batch(self, batch_size, rng):
        A, B = self.datasetA, self.datasetB
        T = self.n_frames
        outs = [np.empty((batch_size, A[0].shape[0], T), dtype=np.float32) for _ in range(4)]
        for b in range(batch_size):
            for src, xo, mo in ((A, outs[0], outs[1]), (B, outs[2], outs[3])):
                utt = src[rng.randint(len(src))]
                lo = rng.randint(utt.shape[1] - T + 1)
                xo[b] = utt[:, lo:lo + T]
                size = rng.randint(0, self.max_mask_len)
                start = rng.randint(0, T - size)
                mo[b] = 1.0
                mo[b, :, start:start + size] = 0.0
        return outs
