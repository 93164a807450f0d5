"""CPU restatement of the on-device minibatch sampler (csrc/sampler_kernels.hip) -- TEST INFRASTRUCTURE ONLY.

The distributions are the reference's (dataset/vc_dataset.py:33-38 uniform utterance with replacement, :44-46 / :59-61
uniform crop, :51-55 / :66-70 mask size ~ U{0..max_mask_len-1}, start ~ U{0..T-size-1}); the random numbers themselves
are a counter-based SplitMix64 stream that is new in this repo (the reference consumes the global numpy RNG, which the
RNG-exact host path ``dataset.vc_dataset.VCDataset`` keeps).  Index work: the HIP kernel must agree BIT-EXACTLY.
"""
import numpy as np

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def draw_u(key, k, n):
    r = splitmix64((key + k) & M64)
    return ((r >> 32) * n) >> 32


def draw_indices(lens_a, lens_b, B, T, max_mask_len, seed, step):
    """-> int32 [B][2][4] = (utterance, crop lo, mask size, mask start) for side 0 (A) and 1 (B)."""
    out = np.zeros((B, 2, 4), dtype=np.int32)
    base = (splitmix64(seed & M64) + step) & M64
    for b in range(B):
        for side, lens in enumerate((lens_a, lens_b)):
            key = splitmix64((splitmix64(base) + 2 * b + side) & M64)
            utt = draw_u(key, 0, len(lens))
            lo = draw_u(key, 1, int(lens[utt]) - T + 1)
            size = draw_u(key, 2, max_mask_len)
            start = draw_u(key, 3, T - size)
            out[b, side] = (utt, lo, size, start)
    return out


def draw_batch(data_a, data_b, B, T, max_mask_len, seed, step):
    """-> (real_A, mask_A, real_B, mask_B) float32 [B,80,T] and the index table, from lists of [80, T_i] arrays."""
    idx = draw_indices([u.shape[1] for u in data_a], [u.shape[1] for u in data_b], B, T, max_mask_len, seed, step)
    outs = [np.empty((B, 80, T), dtype=np.float32) for _ in range(4)]
    for b in range(B):
        for side, data in enumerate((data_a, data_b)):
            utt, lo, size, start = (int(v) for v in idx[b, side])
            outs[2 * side][b] = data[utt][:, lo:lo + T]
            m = np.ones((80, T), dtype=np.float32)
            m[:, start:start + size] = 0.0
            outs[2 * side + 1][b] = m
    return outs, idx
