// On-device input pipeline of the training step: random utterance, random crop, filling-in-frames mask -- one launch per
// minibatch, written straight into the engine's static input buffers.
//
// Replaces, for the default training loop, the reference's host path dataset/vc_dataset.py:19-77 (every __getitem__
// re-shuffles ALL utterance indices and crops / masks EVERY pair: O(N) host work per sample) + the DataLoader collate +
// four H2D copies per iteration (train.py:82-96, 187-190).  Distributions are the reference's:
//   utterance ~ U{0..n-1}   (the reference returns element `index` of a fresh shuffle, i.e. a uniform draw with replacement
//                            vc_dataset.py:33-38 -- A and B shuffled independently)
//   crop lo   ~ U{0..len-T}                 (:44-46, :59-61)
//   mask size ~ U{0..max_mask_len-1}, start ~ U{0..T-size-1}, mask = ones with [start, start+size) zeroed over all 80 bins
//                                           (:51-55, :66-70)
// The generator is counter-based (SplitMix64 of (seed, step, sample, side, draw index)), so a draw is a pure function of
// its coordinates: reproducible, order-independent, restated bit-exactly in oracle/sampler_oracle.py.
//
// Data layout: the utterance bank of a speaker is ONE [80][total_frames] fp32 matrix (all utterances side by side along
// the frame axis) + an int32 offsets table; a crop row is T contiguous floats -> coalesced 256-byte reads.
#include "mcvc_common.h"
#include "sampler.h"
#include "trace.h"

namespace {

__host__ __device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// k-th draw of stream `key`, uniform on {0..n-1} (multiply-shift of the high 32 bits; bias < n / 2^32)
__host__ __device__ __forceinline__ unsigned draw_u(unsigned long long key, unsigned k, unsigned n)
{
    const unsigned long long r = splitmix64(key + k);
    return (unsigned)(((r >> 32) * (unsigned long long)n) >> 32);
}

__global__ void __launch_bounds__(256) draw_batch_kernel(const DrawArgs a)
{
    const int b = blockIdx.x, side = blockIdx.y;
    const unsigned long long key = splitmix64(splitmix64(splitmix64(a.seed) + a.step) + (unsigned long long)(2 * b + side));
    const int utt = (int)draw_u(key, 0, (unsigned)a.n[side]);
    const int o0 = a.offs[side][utt], len = a.offs[side][utt + 1] - o0;
    const int lo = (int)draw_u(key, 1, (unsigned)(len - a.T + 1));
    const int size = (int)draw_u(key, 2, (unsigned)a.max_mask_len);
    const int start = (int)draw_u(key, 3, (unsigned)(a.T - size));
    if (a.draws && threadIdx.x == 0) {
        int* d = a.draws + (b * 2 + side) * 4;
        d[0] = utt; d[1] = lo; d[2] = size; d[3] = start;
    }
    const float* src = a.bank[side] + o0 + lo;
    float* real = a.real[side] + (long long)b * MCVC_SAMPLER_BINS * a.T;
    float* mask = a.mask[side] + (long long)b * MCVC_SAMPLER_BINS * a.T;
    const int total = MCVC_SAMPLER_BINS * a.T;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int m = i / a.T, t = i - m * a.T;
        real[i] = src[(long long)m * a.ld[side] + t];
        mask[i] = (t >= start && t < start + size) ? 0.0f : 1.0f;
    }
}

}  // namespace

int mcvc_draw_batch_launch(const DrawArgs& a, hipStream_t s)
{
    if (a.B < 1 || a.T < 1 || a.max_mask_len < 1 || a.max_mask_len > a.T) return MCVC_ERR_INVALID;
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * 2.0 * a.B * MCVC_SAMPLER_BINS * a.T * 3.0);
    hipLaunchKernelGGL(draw_batch_kernel, dim3((unsigned)a.B, 2), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
