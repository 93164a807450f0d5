// Direct convolution with very few output channels (Cout <= 4, stride 1) on the VALU.
//
// lastConvLayer (128 -> 1, 5x15; model.py:207-211), the data-gradient of conv1 / conv1_gates (256 -> 2, 5x15; :116-126), the
// data-gradient of the discriminator's first conv (256 -> 1, 3x3; :290-295) and its output conv (1024 -> 1, 1x3; :322-327)
// have 1-2 useful rows in a 32-row MFMA tile: on the matrix path they ran at 1-5 TFLOP/s (50-130 us each for 0.1-0.4
// GFLOP).  Here the GEMM shape is dropped altogether: one thread owns 4 adjacent output pixels x all CO output channels,
// the haloed input rows of a few channels sit in LDS, a thread pulls its 4+KW-1 row window into registers with 16-byte LDS
// reads and slides the KW taps over it (60 FMAs per 20 LDS floats at KW=15).  The round's weights are gathered from the
// packed K-major matrix (rows (ci,kh,kw), columns co) into LDS next to the planes and read back as broadcast 16-byte reads
// (a first version fetched them with wave-uniform scalar loads: every tap is its own cache line, and the ~1 us scalar-load
// round trip per kernel row made the kernel 40-75 us).  Input channels are split over workgroups; partial sums leave
// through the same split-K slab / accumulate conventions as conv_direct_kernel (the consumer kernel folds the slabs).
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include <stdlib.h>

namespace {

constexpr int kFewTW = 64;       // tile: 16 threads x 4 pixels wide
constexpr int kFewTH = 16;       //       16 rows
constexpr int kFewCC = 4;        // input channels staged per LDS round

__device__ __attribute__((aligned(16))) float g_few_zero[64];         // zero-initialised: DMA source for out-of-image elements

__device__ __forceinline__ void glds4(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

struct FewArgs {
    const float* x; const float* w; const float* bias;
    float* y; float* y_slabs;
    long long x_sb, x_sc, y_sb, y_sc, slab_stride;
    int x_sh, y_sh, y_sw;
    int Cin, H, W, Cout, OH, OW, KH, pad_h, pad_w;
    int w_cout;                  // row pitch of the packed weight matrix
    int tiles_w, tiles_h;
    int nsplit, ch_per_split, NB;
    int PH, PWp;                 // LDS patch rows / pitch (multiple of 4)
    int accumulate;
};

template <int CO, int KW>
__global__ void __launch_bounds__(256) conv_fewout_kernel(const Twin<FewArgs> tw)
{
    const FewArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [2 buffers][kFewCC][PH][PWp]
    constexpr int NV = (4 + KW - 1 + 3) / 4;                            // float4 reads per row window
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tile = blockIdx.x, split = blockIdx.y / a.NB, n = blockIdx.y - split * a.NB;      // (blockIdx.z selects the network of a grouped launch)
    const int oh0 = (tile / a.tiles_w) * kFewTH, ow0 = (tile % a.tiles_w) * kFewTW;
    const int ih0 = oh0 - a.pad_h, iw0 = ow0 - a.pad_w;
    const int plane = a.PH * a.PWp;
    const int c_begin = split * a.ch_per_split;
    int c_end = c_begin + a.ch_per_split;
    if (c_end > a.Cin) c_end = a.Cin;

    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

    const float* xn = a.x + (long long)n * a.x_sb;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows of the patch that an in-image output row of this tile can touch (small images: far fewer than PH)
    int rows = a.OH - oh0 + a.KH - 1;
    if (rows > a.PH) rows = a.PH;
    const int LX = rows * a.PWp;
    const int chunks = (LX + 63) >> 6;
    const float inv_pitch = 1.0f / (float)a.PWp;
    // Staging by LDS-DMA (global_load_lds, 4 B/lane, no VGPRs, all requests in flight), double-buffered: the planes of
    // round r+1 stream in while round r is being computed; one barrier per round.
    constexpr int KWP = (KW + 3) & ~3;                                  // taps of one kernel row, padded to float4s
    const int wsz = kFewCC * a.KH * CO * KWP;                           // weights of one round: [ci][kh][co][KWP]
    const int bufsz = kFewCC * plane + ((wsz + 3) & ~3);
    auto stage = [&](int c0, float* dst) {
        for (int job = wave; job < kFewCC * chunks; job += 4) {
            const int ci = job / chunks, ch = job - ci * chunks;
            const int idx = ch * 64 + lane;
            int r = (int)((float)idx * inv_pitch);
            int c = idx - r * a.PWp;
            if (c < 0) { c += a.PWp; --r; } else if (c >= a.PWp) { c -= a.PWp; ++r; }
            const int ih = ih0 + r, iw = iw0 + c, cg = c0 + ci;
            const bool ok = (cg < c_end) && (ih >= 0) && (ih < a.H) && (iw >= 0) && (iw < a.W);
            const float* src = ok ? (xn + (long long)cg * a.x_sc + (long long)ih * a.x_sh + iw) : (g_few_zero + lane);
            if (idx < LX) glds4(src, dst + ci * plane + ch * 64);
        }
        float* wdst = dst + kFewCC * plane;
        for (int ch = wave; ch * 64 < wsz; ch += 4) {
            const int d = ch * 64 + lane;
            const int kw = d % KWP; int t = d / KWP;
            const int co = t % CO; t /= CO;
            const int kh = t % a.KH, ci = t / a.KH;
            const bool ok = (d < wsz) && (kw < KW) && (c0 + ci < c_end) && (co < a.Cout);
            const float* src = ok ? (a.w + ((long long)((c0 + ci) * a.KH + kh) * KW + kw) * a.w_cout + co) : (g_few_zero + lane);
            if (d < wsz) glds4(src, wdst + ch * 64);
        }
    };
    const int nrounds = (c_end > c_begin) ? (c_end - c_begin + kFewCC - 1) / kFewCC : 0;
    if (nrounds > 0) stage(c_begin, xs);
    for (int rd = 0; rd < nrounds; ++rd) {
        const int c0 = c_begin + rd * kFewCC;
        float* cur = xs + (rd & 1) * bufsz;
        __syncthreads();                                   // DMA of this round landed (vmcnt(0) in front of the barrier);
                                                           // everyone is done with the other buffer
        if (rd + 1 < nrounds) stage(c0 + kFewCC, xs + ((rd + 1) & 1) * bufsz);
        const float* wcur = cur + kFewCC * plane;
#pragma unroll 1
        for (int ci = 0; ci < kFewCC; ++ci) {
            if (c0 + ci >= c_end) break;
            const float* xrow = cur + ci * plane + ty * a.PWp + tx * 4;
#pragma unroll 1
            for (int kh = 0; kh < a.KH; ++kh) {
                float xr[4 * NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 q = *reinterpret_cast<const float4*>(xrow + kh * a.PWp + 4 * v);
                    xr[4 * v] = q.x; xr[4 * v + 1] = q.y; xr[4 * v + 2] = q.z; xr[4 * v + 3] = q.w;
                }
                const float* wr = wcur + (ci * a.KH + kh) * CO * KWP;       // same address in every lane: LDS broadcast
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    float wv[KWP];
#pragma unroll
                    for (int v = 0; v < KWP / 4; ++v) {
                        const float4 q = *reinterpret_cast<const float4*>(wr + co * KWP + 4 * v);
                        wv[4 * v] = q.x; wv[4 * v + 1] = q.y; wv[4 * v + 2] = q.z; wv[4 * v + 3] = q.w;
                    }
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv[kw], xr[p + kw], acc[co][p]);
                }
            }
        }
    }

    // ---- epilogue: split 0 (+bias) -> y, split s >= 1 -> slab s-1; accumulate mode adds into y
    const int oh = oh0 + ty;
    if (oh >= a.OH) return;
    float* base = (split == 0 || a.accumulate) ? a.y : (a.y_slabs + (long long)(split - 1) * a.slab_stride);
    base += (long long)n * a.y_sb + (long long)oh * a.y_sh;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (co >= a.Cout) break;
        const float b = (a.bias != nullptr && split == 0) ? a.bias[co] : 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ow = ow0 + tx * 4 + p;
            if (ow >= a.OW) continue;
            float* dst = base + (long long)co * a.y_sc + (long long)ow * a.y_sw;
            const float v = acc[co][p] + b;
            if (a.accumulate) { if (a.nsplit > 1) unsafeAtomicAdd(dst, v); else *dst += v; }
            else *dst = v;
        }
    }
}

// ---- the same convolution on the matrix cores (KH x KW = 5 x 15: lastConvLayer forward, conv1's data gradient) --------------------------
// With one or two output channels a (co, pixel) GEMM wastes 30 of 32 rows; here the KERNEL COLUMN is the row dimension instead:
//     Z[kw][h][w'] = sum over (ci, kh) of W[ci][kh][kw] * x[ci][h + kh][w']          16 rows (15 taps + a zero row) x haloed pixels
//     y[h][w]      = sum over kw of Z[kw][h][w + kw]
// i.e. M = 16, K = Cin * KH, N = the haloed row (TW + 14 columns): v_mfma_f32_16x16x4_f32 with the four k of an instruction = four (ci, kh)
// pairs, the B operand read straight from the staged planes (lane n of group k reads x[ci_k][h + kh_k][w0 + n]).  15/16 of the rows and
// 64/80 of the columns are useful: 1.33x the direct FLOPs on a pipe with 4x the VALU kernel's rate.  A wave owns 4 output rows x 5 column
// tiles (20 accumulators per output channel), so one A read serves 20 MFMAs; the anti-diagonal sum runs once per tile through LDS.
// Staging, the split over input channels and the slab / accumulate conventions are those of conv_fewout_kernel.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
constexpr int kFewMfmaPW = 84, kFewMfmaPH = kFewTH + 4;                 // patch pitch / rows for KH x KW = 5 x 15
constexpr int kFewMfmaNCH = (kFewMfmaPH * (kFewMfmaPW / 4) + 63) / 64;  // 16-byte DMA instructions per plane (7)
template <int CO>
__global__ void __launch_bounds__(256, (CO == 1 ? 2 : 1)) conv_fewout_mfma_kernel(const Twin<FewArgs> tw)
{
    const FewArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [2 buffers][kFewCC planes | weights]; reused by the epilogue
    constexpr int KH = 5, KW = 15, KWP = 16, NTI = (kFewTW + KW - 1 + 15) / 16;      // 5 column tiles of 16 haloed pixels
    constexpr int PWp = kFewMfmaPW, PH = kFewMfmaPH, P4 = PWp / 4, plane = PH * PWp, NCH = kFewMfmaNCH;
    static_assert((kFewCC * KH) % 4 == 0 && kFewCC == 4, "k blocks of four; one wave stages one channel");
    const int tid = threadIdx.x;
    const int NB = a.NB, tiles_w = a.tiles_w;
    const int tile = blockIdx.x, split = blockIdx.y / NB, n = blockIdx.y - split * NB;
    const int oh0 = (tile / tiles_w) * kFewTH, ow0 = (tile % tiles_w) * kFewTW;
    const int H = a.H, W = a.W, OH = a.OH, OW = a.OW, x_sh = a.x_sh, w_cout = a.w_cout, Cout = a.Cout;
    const long long x_sc = a.x_sc;
    const int ih0 = oh0 - a.pad_h, iw0 = ow0 - a.pad_w;
    const int c_begin = split * a.ch_per_split;
    int c_end = c_begin + a.ch_per_split;
    if (c_end > a.Cin) c_end = a.Cin;
    const float* xn = a.x + (long long)n * a.x_sb;
    const float* wbase = a.w;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, kq = lane >> 4;
    // The patch is shifted right by `shc` columns so that LDS column 4q holds an image column that is a multiple of 4: whole 16-byte pieces
    // are inside or outside the image, and the planes stream in by 16-byte LDS-DMA (the 4-byte form moves 0.6 dword per clock and CU: 430 of
    // the 516 us of a 64-sample lastConvLayer forward were staging).  Out-of-image pieces are never written: both buffers are cleared once and
    // a tile's halo sits at the same places in every round.  Wave w stages channel w of the round; a piece's image offset is round-invariant.
    const int shc = (4 - (a.pad_w & 3)) & 3;
    int poff[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int idx = ch * 64 + lane;
        const int r = idx / P4, c = idx - r * P4;
        const int ih = ih0 + r, iw = iw0 - shc + 4 * c;
        poff[ch] = (r < PH && ih >= 0 && ih < H && iw >= 0 && iw + 3 < W) ? ih * x_sh + iw : -1;
    }
    constexpr int wsz = kFewCC * KH * CO * KWP;                         // weights of one round: [ci][kh][co][16]
    constexpr int NWD = (wsz + 255) / 256;                              // 4-byte DMA instructions per wave for them
    constexpr int bufsz = kFewCC * plane + wsz;
    int woff[NWD];
#pragma unroll
    for (int i = 0; i < NWD; ++i) {
        const int d = (wave + 4 * i) * 64 + lane;
        const int kw = d % KWP; int t = d / KWP;
        const int co = t % CO; t /= CO;                                 // t = ci * KH + kh
        woff[i] = (d < wsz && kw < KW && co < Cout) ? (t * KW + kw) * w_cout + co : -1;
    }
    for (int i = tid; i < (2 * bufsz) >> 2; i += 256) reinterpret_cast<float4*>(xs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    auto stage = [&](int c0, float* dst) {
        if (c0 + kFewCC > c_end) {                                      // partial last round: its missing channels read as zeros
            for (int i = tid; i < kFewCC * plane; i += 256)
                if (c0 + i / plane >= c_end) dst[i] = 0.f;
            for (int i = tid; i < wsz; i += 256)
                if (c0 + i / (KH * CO * KWP) >= c_end) dst[kFewCC * plane + i] = 0.f;
        }
        if (c0 + wave < c_end) {
            const float* xc = xn + (long long)(c0 + wave) * x_sc;
            float* pd = dst + wave * plane;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (poff[ch] >= 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xc + poff[ch]),
                                                     (__attribute__((address_space(3))) void*)(pd + ch * 256), 16, 0, 0);
        }
        const float* wr = wbase + (long long)c0 * KH * KW * w_cout;
        float* wdst = dst + kFewCC * plane;
#pragma unroll
        for (int i = 0; i < NWD; ++i) {
            const int d0 = (wave + 4 * i) * 64;                         // (rows of channels past c_end stay zero: cleared above)
            if (woff[i] >= 0 && c0 + (d0 + lane) / (KH * CO * KWP) < c_end) glds4(wr + woff[i], wdst + d0);
        }
    };
    f32x4_ acc[CO][4][NTI];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NTI; ++j) acc[co][r][j] = f32x4_{0.f, 0.f, 0.f, 0.f};
    const int nrounds = (c_end > c_begin) ? (c_end - c_begin + kFewCC - 1) / kFewCC : 0;
    if (nrounds > 0) stage(c_begin, xs);
    for (int rd = 0; rd < nrounds; ++rd) {
        float* cur = xs + (rd & 1) * bufsz;
        __syncthreads();
        if (rd + 1 < nrounds) stage(c_begin + (rd + 1) * kFewCC, xs + ((rd + 1) & 1) * bufsz);
        const float* wcur = cur + kFewCC * plane;
        // software pipeline over the five k blocks of the round: the 20 B values (+ A) of block b+1 are read while block b multiplies; the
        // order is pinned (left alone the scheduler pairs every ds_read with its MFMAs: an LDS round trip per two MFMAs)
        constexpr int NBLK = kFewCC * KH / 4;
        float av[2][CO], bv[2][4][NTI];
        auto load_blk = [&](int blk, float (&a_)[CO], float (&b_)[4][NTI]) {
            const int k = 4 * blk + kq;                                 // this lane group's (ci, kh)
            const int ci = k / KH, kh = k - ci * KH;
#pragma unroll
            for (int co = 0; co < CO; ++co) a_[co] = wcur[(k * CO + co) * KWP + ln];
            const float* xb = cur + ci * plane + (4 * wave + kh) * PWp + shc + ln;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NTI; ++j) b_[r][j] = xb[r * PWp + 16 * j];
        };
        load_blk(0, av[0], bv[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * NTI + CO, 0);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            if (blk + 1 < NBLK) load_blk(blk + 1, av[(blk + 1) & 1], bv[(blk + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NTI; ++j)
#pragma unroll
                    for (int co = 0; co < CO; ++co)
                        acc[co][r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[blk & 1][co], bv[blk & 1][r][j], acc[co][r][j], 0, 0, 0);
            if (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, 4 * NTI + CO, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTI * CO, 0);
        }
    }
    // ---- anti-diagonal sums through LDS (per wave: [CO][16][NTI * 16]), then the epilogue of conv_fewout_kernel
    __syncthreads();
    constexpr int ZW = NTI * 16;
    float* zs = xs + wave * (CO * 16 * ZW);
    const int nsplit = a.nsplit, accumulate = a.accumulate, y_sh = a.y_sh, y_sw = a.y_sw;
    const long long y_sc = a.y_sc;
    const float* bias = a.bias;
    float* base = (split == 0 || accumulate) ? a.y : (a.y_slabs + (long long)(split - 1) * a.slab_stride);
    base += (long long)n * a.y_sb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int co = 0; co < CO; ++co)
#pragma unroll
            for (int j = 0; j < NTI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) zs[(co * 16 + 4 * kq + q) * ZW + 16 * j + ln] = acc[co][r][j][q];
        __syncthreads();
        const int oh = oh0 + 4 * wave + r, ow = ow0 + lane;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float v = 0.f;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) v += zs[(co * 16 + kw) * ZW + lane + kw];
            if (co < Cout && oh < OH && ow < OW) {
                if (bias != nullptr && split == 0) v += bias[co];
                float* dst = base + (long long)oh * y_sh + (long long)co * y_sc + (long long)ow * y_sw;
                if (accumulate) { if (nsplit > 1) unsafeAtomicAdd(dst, v); else *dst += v; }
                else *dst = v;
            }
        }
        __syncthreads();
    }
}

// ---- the discriminators' output layer (1 x 3 conv, C channels -> 1, padding (0,1), + sigmoid; model.py:322-327, 348) -------------------
// 80 output values per sample from C = 1024 channels: as a split-K job of the kernel above it took 21 us + 17 us for the consumer that folded
// 64 slabs.  One wave per output value instead: the lanes stride over the channels (all loads independent), a butterfly sums them, lane 0
// stores the logit (+ bias) and its sigmoid.  w = the OIHW parameter itself ([1][C][1][3]).
struct DiscOutFwdKArgs { const float* x; const float* w; const float* bias; float* logit; float* out; int NB; int C; int H; int W; };
__global__ void __launch_bounds__(256) disc_out_fwd_kernel(const Twin<DiscOutFwdKArgs> tw)
{
    const DiscOutFwdKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ x = ka_.x;
    const float* __restrict__ w = ka_.w;
    const float* __restrict__ bias = ka_.bias;
    float* __restrict__ logit = ka_.logit;
    float* __restrict__ out = ka_.out;
    int NB = ka_.NB;
    int C = ka_.C;
    int H = ka_.H;
    int W = ka_.W;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int HW = H * W;
    if (o >= NB * HW) return;
    const int n = o / HW, p = o - n * HW, wc = p % W;
    const float* xp = x + (long long)n * C * HW + p;
    const bool hasl = wc > 0, hasr = wc + 1 < W;
    float acc = 0.f;
#pragma unroll 4
    for (int c = lane; c < C; c += 64) {
        const float* xc = xp + (long long)c * HW;
        const float x0 = hasl ? xc[-1] : 0.f, x1 = xc[0], x2 = hasr ? xc[1] : 0.f;
        acc = fmaf(w[3 * c], x0, fmaf(w[3 * c + 1], x1, fmaf(w[3 * c + 2], x2, acc)));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) {
        const float v = acc + bias[0];
        logit[o] = v;
        out[o] = 1.0f / (1.0f + __expf(-v));
    }
}

// its data-gradient: dx[n][c][h][w] = sum_kw w[c][kw] * dlogit[n][h][w + 1 - kw]
struct DiscOutDgradKArgs { const float* dl; const float* w; float* dx; int NB; int C; int H; int W; };
__global__ void __launch_bounds__(256) disc_out_dgrad_kernel(const Twin<DiscOutDgradKArgs> tw)
{
    const DiscOutDgradKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ dl = ka_.dl;
    const float* __restrict__ w = ka_.w;
    float* __restrict__ dx = ka_.dx;
    int NB = ka_.NB;
    int C = ka_.C;
    int H = ka_.H;
    int W = ka_.W;
    const long long total = (long long)NB * C * H * W;
    const int HW = H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const long long nc = i / HW;
        const int c = (int)(nc % C), n = (int)(nc / C);
        const int wc = p % W;
        const float* d = dl + (long long)n * HW + p;
        float v = w[3 * c + 1] * d[0];
        if (wc + 1 < W) v = fmaf(w[3 * c], d[1], v);
        if (wc > 0) v = fmaf(w[3 * c + 2], d[-1], v);
        dx[i] = v;
    }
}

// ---- the discriminators' first layer: 3x3 conv from ONE channel + x*sigmoid(x) (model.py:290-295, 343-344) -----------------------------
// 9 multiplies per output: on the matrix kernel + a separate activation pass it was 12 + 7 us.  A thread keeps the 3x3 neighbourhood of its
// pixel in registers and walks COB output channels (weights: wave-uniform scalar loads); it stores the pre-activation (backward) and the
// activation, both coalesced along the pixels.  w = the OIHW parameter ([Cout][1][3][3]).
constexpr int kDiscC1Cob = 16;
struct DiscConv1FwdKArgs { const float* x; const float* w; const float* bias; float* c0; float* y0; int Cout; int H; int W; int xs; };
__global__ void __launch_bounds__(256) disc_conv1_fwd_kernel(const Twin<DiscConv1FwdKArgs> tw)
{
    const DiscConv1FwdKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ x = ka_.x;
    const float* __restrict__ w = ka_.w;
    const float* __restrict__ bias = ka_.bias;
    float* __restrict__ c0 = ka_.c0;
    float* __restrict__ y0 = ka_.y0;
    int Cout = ka_.Cout;
    int H = ka_.H;
    int W = ka_.W;
    const int HW = H * W;
    const int cbs = (Cout + kDiscC1Cob - 1) / kDiscC1Cob;          // (blockIdx.z selects the network of a grouped launch: the sample rides in y)
    const int p = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y / cbs, co0 = (blockIdx.y - n * cbs) * kDiscC1Cob;
    if (p >= HW) return;
    const int h = p / W, wc = p - h * W;
    const float* xp = x + (long long)n * HW;
    float xv[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = h + kh - 1, iw = wc + kw - 1;
            xv[kh * 3 + kw] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? xp[ih * W + iw] : 0.f;
        }
    const long long o0 = ((long long)n * Cout + co0) * HW + p;
    // xs: the activation goes out in the phase-split padded layout the implicit-GEMM forward of downSample1 gathers from (sgemm.h);
    // the thread of pixel (h, wc) also writes the zero borders next to it (row 0 of its sub-plane for h < 2, columns 0..3 for wc < 2)
    const int xs = ka_.xs;
    const int pw = W / 2 + 4;
    const long long plane = (long long)(H / 2 + 1) * pw;
    const long long x0 = ((long long)n * Cout + co0) * 4 * plane + (long long)((h & 1) * 2 + (wc & 1)) * plane;
    const long long xo = x0 + (long long)((h >> 1) + 1) * pw + (wc >> 1) + 4;
#pragma unroll 4
    for (int j = 0; j < kDiscC1Cob; ++j) {
        if (co0 + j >= Cout) break;
        const float* wr = w + (co0 + j) * 9;
        float v = bias[co0 + j];
#pragma unroll
        for (int t = 0; t < 9; ++t) v = fmaf(wr[t], xv[t], v);
        c0[o0 + (long long)j * HW] = v;
        const float yv = v / (1.0f + __expf(-v));
        if (!xs) { y0[o0 + (long long)j * HW] = yv; continue; }
        float* yp = y0 + (long long)j * 4 * plane;
        yp[xo] = yv;
        if (h < 2) yp[x0 + (wc >> 1) + 4] = 0.f;
        if (wc < 2) {
            float* r = yp + x0 + (long long)((h >> 1) + 1) * pw;
            r[0] = 0.f; r[1] = 0.f; r[2] = 0.f; r[3] = 0.f;
            if (h < 2) { float* r0 = yp + x0; r0[0] = 0.f; r0[1] = 0.f; r0[2] = 0.f; r0[3] = 0.f; }
        }
    }
}

static int few_enabled()
{
    static const int v = mcvc_knob("MCVC_FEWOUT", 1);
    return v;
}

static int few_nsplit(const ConvProblem& p, int NB, int allow_split)
{
    if (!allow_split) return 1;
    const int rounds = cdiv_i(p.Cin, kFewCC);
    const int tiles = cdiv_i(p.OW, kFewTW) * cdiv_i(p.OH, kFewTH) * NB;
    int ns = cdiv_i(512, tiles);                     // enough workgroups for two per CU
    if (ns > rounds) ns = rounds;
    if (ns > 64) ns = 64;                            // every slab is one more read of the (tiny) output by the consumer
    if (ns < 1) ns = 1;
    return ns;
}

template <int CO, int KW>
static int few_launch_t(const FewArgs& a, dim3 grid, size_t lds, hipStream_t s)
{
    mcvc_launch((conv_fewout_kernel<CO, KW>), grid, dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

int mcvc_disc_out_fwd_launch(const float* x, const float* w, const float* bias, float* logit, float* out, int NB, int C, int H, int W, hipStream_t s)
{
    const int outs = NB * H * W;
    TraceScope ts(K_CONV_FEW, s, 2.0 * 3 * C * outs, 4.0 * ((double)outs * C + 2.0 * outs));
    mcvc_launch(disc_out_fwd_kernel, dim3((unsigned)cdiv_i(outs, 4)), dim3(256), 0, s, DiscOutFwdKArgs{x, w, bias, logit, out, NB, C, H, W});
    return (int)hipGetLastError();
}

int mcvc_disc_out_dgrad_launch(const float* dlogit, const float* w, float* dx, int NB, int C, int H, int W, hipStream_t s)
{
    const long long total = (long long)NB * C * H * W;
    TraceScope ts(K_CONV_FEW, s, 2.0 * 3 * (double)total, 4.0 * (double)total);
    long long blocks = cdiv_ll(total, 256);
    if (blocks > 4096) blocks = 4096;
    mcvc_launch(disc_out_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, DiscOutDgradKArgs{dlogit, w, dx, NB, C, H, W});
    return (int)hipGetLastError();
}

int mcvc_disc_conv1_fwd_launch(const float* x, const float* w, const float* bias, float* c0, float* y0, int NB, int Cout, int H, int W, hipStream_t s, int xs)
{
    const double outs = (double)NB * Cout * H * W;
    TraceScope ts(K_CONV_FEW, s, 2.0 * 9 * outs, 4.0 * (2.0 * outs + (double)NB * H * W));
    mcvc_launch(disc_conv1_fwd_kernel, dim3((unsigned)cdiv_i(H * W, 256), (unsigned)(cdiv_i(Cout, kDiscC1Cob) * NB)), dim3(256), 0, s, DiscConv1FwdKArgs{x, w, bias, c0, y0, Cout, H, W, xs});
    return (int)hipGetLastError();
}

bool mcvc_fewout_applies(const ConvProblem& p)
{
    if (!few_enabled()) return false;
    if (p.Cout > 4 || p.stride != 1) return false;
    if (p.KW != 3 && p.KW != 15) return false;
    const int PH = kFewTH + p.KH - 1, PWp = round_up_i(kFewTW + p.KW - 1 + 3, 4);
    return (size_t)2 * (kFewCC * PH * PWp + kFewCC * p.KH * 4 * ((p.KW + 3) & ~3) + 4) * sizeof(float) <= 64 * 1024;
}

int mcvc_fewout_plan_nsplit(const ConvProblem& p, int NB, int allow_split) { return few_nsplit(p, NB, allow_split); }

int mcvc_fewout_launch(const ConvProblem& p, int NB, const ConvIO& io, const float* wpk, int w_cout, const float* bias, hipStream_t s)
{
    if (!mcvc_fewout_applies(p) || io.shuffle || io.nsplit < 1) return MCVC_ERR_INVALID;
    FewArgs a{};
    a.x = io.x; a.x_sb = io.x_sb; a.x_sc = io.x_sc; a.x_sh = io.x_sh;
    a.y = io.y; a.y_sb = io.y_sb; a.y_sc = io.y_sc; a.y_sh = io.y_sh; a.y_sw = io.y_sw;
    a.y_slabs = io.slabs; a.slab_stride = io.slab_stride;
    a.w = wpk; a.w_cout = w_cout; a.bias = bias;
    a.Cin = p.Cin; a.H = p.H; a.W = p.W; a.Cout = p.Cout; a.OH = p.OH; a.OW = p.OW; a.KH = p.KH; a.pad_h = p.pad_h; a.pad_w = p.pad_w;
    a.tiles_w = cdiv_i(p.OW, kFewTW); a.tiles_h = cdiv_i(p.OH, kFewTH);
    a.nsplit = io.nsplit; a.NB = NB;
    a.ch_per_split = round_up_i(cdiv_i(p.Cin, io.nsplit), 1);
    a.PH = kFewTH + p.KH - 1;
    a.PWp = round_up_i(kFewTW + p.KW - 1 + 3, 4);           // (+3: the last float4 of a row window may read past PW)
    a.accumulate = io.accumulate;
    dim3 grid((unsigned)(a.tiles_w * a.tiles_h), (unsigned)(NB * io.nsplit));
    const int co_t = p.Cout <= 1 ? 1 : (p.Cout <= 2 ? 2 : 4);
    const size_t lds = (size_t)2 * (kFewCC * a.PH * a.PWp + kFewCC * p.KH * co_t * ((p.KW + 3) & ~3) + 4) * sizeof(float);
    const double px = (double)NB * p.OH * p.OW;
    TraceScope ts(K_CONV_FEW, s, 2.0 * px * p.Cout * p.Cin * p.KH * p.KW,
                  4.0 * ((double)NB * p.Cin * p.H * p.W + (double)p.Cin * p.KH * p.KW * p.Cout + px * p.Cout * io.nsplit));
    const int co = co_t;
    // 5 x 15 kernels with one or two output channels: the matrix-core form (MCVC_FEWOUT_MFMA=0: the VALU kernel)
    static const int mfma = mcvc_knob("MCVC_FEWOUT_MFMA", 1);
    const bool dma16 = ((a.x_sh & 3) == 0) && ((p.W & 3) == 0) && ((a.x_sc & 3) == 0) && ((a.x_sb & 3) == 0) &&
                       ((reinterpret_cast<unsigned long long>(a.x) & 15ull) == 0) && (long long)p.H * a.x_sh < (1LL << 30);
    if (mfma && p.KW == 15 && p.KH == 5 && co <= 2 && dma16 && a.PWp == kFewMfmaPW && a.PH == kFewMfmaPH) {
        const size_t stg = (size_t)2 * (kFewCC * kFewMfmaPH * kFewMfmaPW + kFewCC * 5 * co * 16) * sizeof(float);
        const size_t epi = (size_t)4 * co * 16 * 80 * sizeof(float);
        const size_t need = stg > epi ? stg : epi;
        if (co == 1) mcvc_launch(conv_fewout_mfma_kernel<1>, grid, dim3(256), need, s, a);
        else mcvc_launch(conv_fewout_mfma_kernel<2>, grid, dim3(256), need, s, a);
        return (int)hipGetLastError();
    }
    if (p.KW == 15) return co == 1 ? few_launch_t<1, 15>(a, grid, lds, s) : co == 2 ? few_launch_t<2, 15>(a, grid, lds, s) : few_launch_t<4, 15>(a, grid, lds, s);
    return co == 1 ? few_launch_t<1, 3>(a, grid, lds, s) : co == 2 ? few_launch_t<2, 3>(a, grid, lds, s) : few_launch_t<4, 3>(a, grid, lds, s);
}

// ---- weight gradient of a convolution with ONE output channel (stride 1) ----------------------------------------------------------
// lastConvLayer (128 -> 1, 5x15) and the discriminator's output conv (1024 -> 1, 1x3): dW[ci][kh][kw] += sum over samples and pixels of
// dy[oh][ow] * x[ci][oh + kh - ph][ow + kw - pw].  On the MFMA weight-gradient kernel one of 32 rows did useful work (66-80 us per launch
// at bs=1, 2 ms at 64 samples).  Here a workgroup owns one input channel (and a chunk of the samples): the dy plane and the haloed x plane
// sit in LDS, thread (tap, row group) slides its tap over its rows -- dy reads are broadcasts, x reads walk consecutive columns -- and
// the row groups are summed through LDS.  With more than one sample chunk the result is added with atomics (one chunk in deterministic mode).
namespace {

struct FewWgradArgs {
    const float* x; long long x_sn, x_sc; int x_sh;
    const float* dy; long long dy_sn; int dy_sh;
    float* dw;                    // [Cin][KH][KW]
    int N, Cin, H, W, KH, KW, ph, pw;
    int XW;                       // LDS pitch of the haloed x plane
    int band;                     // output rows per (sample, band) unit
};

__global__ void __launch_bounds__(256) wgrad_cout1_kernel(const Twin<FewWgradArgs> tw)
{
    const FewWgradArgs a = tw.v[blockIdx.z];
    extern __shared__ float sm[];
    const int BH = a.band;                       // output rows per unit; units = (sample, band), shared out over gridDim.y workgroups
    const int XH = BH + a.KH - 1;
    float* xs = sm;                              // [XH][XW]
    float* dys = sm + XH * a.XW;                 // [BH][W]
    float* red = dys + BH * a.W;                 // [groups][taps]
    const int tid = threadIdx.x;
    const int ci = blockIdx.x;
    const int ntap = a.KH * a.KW;
    const int ngr = 256 / ntap;
    const int tap = tid % ntap, rg = tid / ntap;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int nbands = (a.H + BH - 1) / BH;
    const int units = a.N * nbands;
    float acc = 0.f;
    for (int u = blockIdx.y; u < units; u += gridDim.y) {
        const int n = u / nbands, h0 = (u - n * nbands) * BH;
        const int rows = (a.H - h0 < BH) ? a.H - h0 : BH;
        const float* xp = a.x + (long long)n * a.x_sn + (long long)ci * a.x_sc;
        __syncthreads();
        for (int i = tid; i < (rows + a.KH - 1) * a.XW; i += 256) {
            const int r = i / a.XW, c = i - r * a.XW;
            const int ih = h0 + r - a.ph, iw = c - a.pw;
            xs[i] = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? xp[(long long)ih * a.x_sh + iw] : 0.f;
        }
        const float* dp = a.dy + (long long)n * a.dy_sn + (long long)h0 * a.dy_sh;
        for (int i = tid; i < rows * a.W; i += 256) { const int r = i / a.W, c = i - r * a.W; dys[i] = dp[(long long)r * a.dy_sh + c]; }
        __syncthreads();
        if (rg < ngr) {
            for (int oh = rg; oh < rows; oh += ngr) {
                const float* d = dys + oh * a.W;
                const float* xr = xs + (oh + kh) * a.XW + kw;
                float s0 = 0.f, s1 = 0.f;
                int ow = 0;
                for (; ow + 1 < a.W; ow += 2) { s0 += d[ow] * xr[ow]; s1 += d[ow + 1] * xr[ow + 1]; }
                if (ow < a.W) s0 += d[ow] * xr[ow];
                acc += s0 + s1;
            }
        }
    }
    __syncthreads();
    if (rg < ngr) red[rg * ntap + tap] = acc;
    __syncthreads();
    if (tid < ntap && (int)blockIdx.y < units) {
        float s = 0.f;
        for (int g = 0; g < ngr; ++g) s += red[g * ntap + tid];
        float* d = a.dw + (long long)ci * ntap + tid;
        if (gridDim.y > 1) unsafeAtomicAdd(d, s); else *d += s;
    }
}

// One INPUT channel, 3x3, stride 1, padding 1 (the discriminators' first conv, model.py:290-295): dw[co][kh][kw] = sum over samples and
// pixels of dy[n][co][h][w] * x[n][h+kh-1][w+kw-1].  The matrix kernel ran this 9-column problem at 0.6 TF/s (36 us, last launch of every
// discriminator backward pass).  Here a workgroup owns one output channel and a share of (sample, band of rows) units: the haloed band of x
// sits in LDS, a thread multiplies its dy values with the nine neighbours, the nine sums are folded over the workgroup.
constexpr int kCin1Band = 20;
struct WgradCin1KArgs { const float* x; long long x_sn; int x_sh; const float* dy; long long dy_sn; long long dy_sc; int dy_sh; float* dw; int NB; int H; int W; };
__global__ void __launch_bounds__(256) wgrad_cin1_kernel(const Twin<WgradCin1KArgs> tw)
{
    const WgradCin1KArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ x = ka_.x;
    long long x_sn = ka_.x_sn;
    int x_sh = ka_.x_sh;
    const float* __restrict__ dy = ka_.dy;
    long long dy_sn = ka_.dy_sn;
    long long dy_sc = ka_.dy_sc;
    int dy_sh = ka_.dy_sh;
    float* __restrict__ dw = ka_.dw;
    int NB = ka_.NB;
    int H = ka_.H;
    int W = ka_.W;
    extern __shared__ float sm[];
    const int XW = W + 2;
    float* xs = sm;                                   // [kCin1Band + 2][XW]
    float* red = sm + (kCin1Band + 2) * XW;           // [4][9]
    const int tid = threadIdx.x, co = blockIdx.x;
    const int bands = (H + kCin1Band - 1) / kCin1Band;
    const int units = NB * bands;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int u = blockIdx.y; u < units; u += gridDim.y) {
        const int n = u / bands, h0 = (u - n * bands) * kCin1Band;
        const int rows = (H - h0 < kCin1Band) ? H - h0 : kCin1Band;
        const float* xp = x + (long long)n * x_sn;
        __syncthreads();
        for (int i = tid; i < (rows + 2) * XW; i += 256) {
            const int r = i / XW, c = i - r * XW;
            const int ih = h0 + r - 1, iw = c - 1;
            xs[i] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? xp[(long long)ih * x_sh + iw] : 0.f;
        }
        __syncthreads();
        const float* dp = dy + (long long)n * dy_sn + (long long)co * dy_sc + (long long)h0 * dy_sh;
        for (int i = tid; i < rows * W; i += 256) {
            const int r = i / W, c = i - r * W;
            const float d = dp[(long long)r * dy_sh + c];
            const float* xr = xs + r * XW + c;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(d, xr[kh * XW + kw], acc[kh * 3 + kw]);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[t] += __shfl_xor(acc[t], o, 64);
    }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[(tid >> 6) * 9 + t] = acc[t];
    }
    __syncthreads();
    if (tid < 9 && (int)blockIdx.y < units) {
        const float v = (red[tid] + red[9 + tid]) + (red[18 + tid] + red[27 + tid]);
        float* d = dw + co * 9 + tid;
        if (gridDim.y > 1) unsafeAtomicAdd(d, v); else *d += v;
    }
}

}  // namespace

namespace {

// ---- lastConvLayer's weight gradient on the matrix cores (KH x KW = 5 x 15, one output channel) -----------------------------------------
//     dW[ci][kh][kw] = sum over (n, h, w) of dy[h][w] * x[ci][h + kh - 2][w + kw - 7]
// With the KERNEL COLUMN as the row dimension and the INPUT CHANNEL as the column dimension this is a GEMM whose A operand is a Toeplitz
// matrix of ONE dy row, shared by every channel and kernel row:
//     Z(h, kh)[kw][ci] = sum over w' of  dy[h][w' - kw] * x[ci][h + kh - 2][w' - 7]          M = 16 (15 + a zero row), N = 16 channels, K = W + 14
// v_mfma_f32_16x16x4_f32: lane (m = kw, k) of A reads the zero-padded dy row at w' - kw, lane (k, n = ci) of B reads the staged plane of
// channel ci.  A workgroup owns 16 channels x a band of 8 rows of one sample; its four waves take two rows each and keep five accumulators
// (one per kernel row), which are summed through LDS and added to dW (atomics when several workgroups share a channel tile).
constexpr int kW1Ch = 16, kW1PW = 84;
template <int BAND> constexpr int kW1PlaneOf = (BAND + 4) * kW1PW + 4;                      // (+4: planes 20 banks apart, rows stay 16-byte aligned)
constexpr int kW1DyW = 64 + 32;                                                             // a dy row with 16 zeros on either side
struct W1Args {
    const float* x; long long x_sn, x_sc; int x_sh;
    const float* dy; long long dy_sn; int dy_sh;
    float* dw;
    int N, Cin, H, W, bands, units, atomic;
};
template <int kW1Band>                                 // rows per (sample, band) unit: 8, or 4 when 8 would leave CUs without a workgroup
__global__ void __launch_bounds__(256, 2) wgrad_cout1_mfma_kernel(const Twin<W1Args> tw)
{
    constexpr int kW1Plane = kW1PlaneOf<kW1Band>;
    const W1Args a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                                   // [16 channels][kW1Plane]: rows h0-2 .. h0+band+1, image column c at LDS column c + 8
    float* dys = sm + kW1Ch * kW1Plane;               // [band][kW1DyW]: dy[h][w] at column w + 16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, kq = lane >> 4;
    const int c0 = blockIdx.x * kW1Ch;
    typedef float f32x4w __attribute__((ext_vector_type(4)));
    f32x4w acc[5];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) acc[kh] = f32x4w{0.f, 0.f, 0.f, 0.f};
    constexpr int P4 = kW1PW / 4;
    for (int u = blockIdx.y; u < a.units; u += gridDim.y) {
        const int n = u / a.bands, h0 = (u - n * a.bands) * kW1Band;
        __syncthreads();
        // stage: 16-byte pieces; whole pieces are inside or outside the image (W % 4 == 0, columns shifted by 8)
        {   // (all of a thread's loads in flight before the first LDS store)
            constexpr int TOT = kW1Ch * (kW1Band + 4) * P4, NIT = (TOT + 255) / 256;
            float4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                const int ci = i / ((kW1Band + 4) * P4), rem = i - ci * ((kW1Band + 4) * P4);
                const int r = rem / P4, c4 = rem - r * P4;
                const int ih = h0 - 2 + r, iw = 4 * c4 - 8;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < TOT && c0 + ci < a.Cin && ih >= 0 && ih < a.H && iw >= 0 && iw + 3 < a.W)
                    v[it] = *reinterpret_cast<const float4*>(a.x + (long long)n * a.x_sn + (long long)(c0 + ci) * a.x_sc + (long long)ih * a.x_sh + iw);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                const int ci = i / ((kW1Band + 4) * P4), rem = i - ci * ((kW1Band + 4) * P4);
                const int r = rem / P4, c4 = rem - r * P4;
                if (i < TOT) *reinterpret_cast<float4*>(xs + ci * kW1Plane + r * kW1PW + 4 * c4) = v[it];
            }
        }
        for (int i = tid; i < kW1Band * (kW1DyW / 4); i += 256) {
            const int r = i / (kW1DyW / 4), c4 = i - r * (kW1DyW / 4);
            const int h = h0 + r, w = 4 * c4 - 16;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < a.H && w >= 0 && w + 3 < a.W) v = *reinterpret_cast<const float4*>(a.dy + (long long)n * a.dy_sn + (long long)h * a.dy_sh + w);
            *reinterpret_cast<float4*>(dys + r * kW1DyW + 4 * c4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < kW1Band / 4; ++rr) {
            const int r = (kW1Band / 4) * wave + rr;                   // output row h0 + r
            // A[m = ln][k]: dy[h][w' - 7 - m + 7 ... ]: with w' = 4 kb + kq the image column of x is w' - 7, and the tap kw = m pairs it with
            // dy column (w' - 7) - (m - 7) = w' - m: LDS column w' - m + 16
            const float* dr = dys + r * kW1DyW + 16 - ln + kq;
            const float* xr = xs + ln * kW1Plane + r * kW1PW + 1 + kq;          // x[ci = ln][h0 - 2 + r + kh][w' - 7] at LDS column w' + 1
            // (software pipeline as in conv_fewout_mfma_kernel: the six operands of block kb + 1 are read while block kb multiplies)
            float av[2], bw[2][5];
            av[0] = dr[0];
#pragma unroll
            for (int kh = 0; kh < 5; ++kh) bw[0][kh] = xr[kh * kW1PW];
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int kb = 0; kb < 20; ++kb) {                          // w' = 0 .. 79 (64 + 14 columns, padded to 80)
                if (kb + 1 < 20) {
                    av[(kb + 1) & 1] = dr[4 * (kb + 1)];
#pragma unroll
                    for (int kh = 0; kh < 5; ++kh) bw[(kb + 1) & 1][kh] = xr[kh * kW1PW + 4 * (kb + 1)];
                }
#pragma unroll
                for (int kh = 0; kh < 5; ++kh)
                    acc[kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb & 1], bw[kb & 1][kh], acc[kh], 0, 0, 0);
                if (kb + 1 < 20) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
            }
        }
    }
    // ---- the four waves' partial sums through LDS: red[wave][kh][kw = 4 kq + q][ci = ln]
    __syncthreads();
    float* red = sm;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[((wave * 5 + kh) * 16 + 4 * kq + q) * 16 + ln] = acc[kh][q];
    __syncthreads();
    for (int i = tid; i < 5 * 15 * kW1Ch; i += 256) {
        const int ci = i / 75, t = i - ci * 75, kh = t / 15, kw = t - kh * 15;
        if (c0 + ci >= a.Cin) continue;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) v += red[((wv * 5 + kh) * 16 + kw) * 16 + ci];
        float* d = a.dw + (long long)(c0 + ci) * 75 + t;
        if (a.atomic) unsafeAtomicAdd(d, v); else *d += v;
    }
}

}  // namespace

namespace {

// ---- conv1's weight gradient on the matrix cores (Cin <= 2 input channels, KH x KW = 5 x 15) ----------------------------------------------
//     dW[co][ci][kh][kw] = sum over (n, h, w) of dy[co][h][w] * x[ci][h + kh - 2][w + kw - 7]
// The kernel column is the row dimension again, now with the OUTPUT channel as the column dimension; the Toeplitz operand comes from x:
//     Z(h, ci, kh)[kw][co] = sum over w of  x[ci][h + kh - 2][w + kw - 7] * dy[co][h][w]            M = 16, N = 16 channels, K = W
// A workgroup owns 64 output channels (16 per wave) and walks (sample, band of 4 rows) units; a wave keeps Cin * 5 accumulators.  One B
// read (dy) serves the Cin * 5 MFMAs of a k block, every A read is 16 consecutive columns of a staged x row.
constexpr int kW2Co = 64;
template <int BAND> constexpr int kW2DyPitch = BAND * 64 + 4;           // dy pitch per channel: 4 banks apart (2-way conflicts at most)
constexpr int kW2XP = 84;
struct W2Args {
    const float* x; long long x_sn, x_sc; int x_sh;
    const float* dy; long long dy_sn, dy_sc; int dy_sh;
    float* dw;
    int N, Cin, Cout, H, W, bands, units, atomic;
};
template <int CI, int kW2Band>
__global__ void __launch_bounds__(256, 2) wgrad_cin2_mfma_kernel(const Twin<W2Args> tw)
{
    constexpr int kW2DyP = kW2DyPitch<kW2Band>;
    const W2Args a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dys = sm;                                      // [64 channels][kW2DyP]: dy[co][h0 + r][w] at r * 64 + w
    float* xs = sm + kW2Co * kW2DyP;                      // [CI][band + 4][kW2XP]: rows h0 - 2 .., image column c at LDS column c + 8
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.x * kW2Co;
    typedef float f32x4w __attribute__((ext_vector_type(4)));
    f32x4w acc[CI][5];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) acc[ci][kh] = f32x4w{0.f, 0.f, 0.f, 0.f};
    constexpr int P4 = kW2XP / 4, XR = kW2Band + 4;
    for (int u = blockIdx.y; u < a.units; u += gridDim.y) {
        const int n = u / a.bands, h0 = (u - n * a.bands) * kW2Band;
        __syncthreads();
        {   // dy: 16-byte pieces, rows of 64; all of a thread's loads are in flight before the first LDS store (a plain load -> store loop
            // serialises 16 global round trips per unit: it was 2/3 of the kernel)
            constexpr int NIT = kW2Co * kW2Band * 16 / 256;
            float4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                const int co = i / (kW2Band * 16), rem = i - co * (kW2Band * 16);
                const int r = rem >> 4, c4 = rem & 15;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co0 + co < a.Cout && h0 + r < a.H)
                    v[it] = *reinterpret_cast<const float4*>(a.dy + (long long)n * a.dy_sn + (long long)(co0 + co) * a.dy_sc + (long long)(h0 + r) * a.dy_sh + 4 * c4);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                const int co = i / (kW2Band * 16), rem = i - co * (kW2Band * 16);
                const int r = rem >> 4, c4 = rem & 15;
                *reinterpret_cast<float4*>(dys + co * kW2DyP + r * 64 + 4 * c4) = v[it];
            }
        }
        for (int i = tid; i < CI * XR * P4; i += 256) {
            const int ci = i / (XR * P4), rem = i - ci * (XR * P4);
            const int r = rem / P4, c4 = rem - r * P4;
            const int ih = h0 - 2 + r, iw = 4 * c4 - 8;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < a.Cin && ih >= 0 && ih < a.H && iw >= 0 && iw + 3 < a.W)
                v = *reinterpret_cast<const float4*>(a.x + (long long)n * a.x_sn + (long long)ci * a.x_sc + (long long)ih * a.x_sh + iw);
            *reinterpret_cast<float4*>(xs + (ci * XR + r) * kW2XP + 4 * c4) = v;
        }
        __syncthreads();
        const float* br = dys + (wave * 16 + ln) * kW2DyP + kq;           // B[k = kq][n = ln]: dy[co][h][w = 4 kb + kq]
        const float* ar = xs + 1 + ln + kq;                               // A[m = ln][k = kq]: x[ci][h + kh - 2][w + kw - 7] at column w + kw + 1
        // software pipeline over the k blocks: the CI * 5 + 1 operands of block kb + 1 are read while block kb multiplies (pinned order; left
        // alone every MFMA waits for its own LDS round trip)
        auto load_kb = [&](int r, int kb, float (&av)[CI][5], float& bv) {
            bv = br[r * 64 + 4 * kb];
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int kh = 0; kh < 5; ++kh) av[ci][kh] = ar[(ci * XR + r + kh) * kW2XP + 4 * kb];
        };
#pragma unroll 1
        for (int r = 0; r < kW2Band; ++r) {
            float av[2][CI][5], bv[2];
            load_kb(r, 0, av[0], bv[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, CI * 5 + 1, 0);
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                if (kb + 1 < 16) load_kb(r, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
#pragma unroll
                for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                    for (int kh = 0; kh < 5; ++kh)
                        acc[ci][kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb & 1][ci][kh], bv[kb & 1], acc[ci][kh], 0, 0, 0);
                if (kb + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, CI * 5 + 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CI * 5, 0);
            }
        }
    }
    // ---- write-out through LDS so that consecutive lanes add to consecutive addresses (the block's [64][Cin * 75] slice of dW is contiguous):
    // acc register q of lane (ln, kq) is (kw = 4 kq + q, co = 16 wave + ln)
    __syncthreads();
    float* red = sm;                                      // [64][Cin * 75]
    const int per = a.Cin * 75;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        if (ci >= a.Cin) break;
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kw = 4 * kq + q;
                if (kw < 15) red[(wave * 16 + ln) * per + (ci * 5 + kh) * 15 + kw] = acc[ci][kh][q];
            }
    }
    __syncthreads();
    int nco = a.Cout - co0; if (nco > kW2Co) nco = kW2Co;
    float* d0 = a.dw + (long long)co0 * per;
    for (int i = tid; i < nco * per; i += 256) {
        if (a.atomic) unsafeAtomicAdd(d0 + i, red[i]); else d0[i] += red[i];
    }
}

}  // namespace

bool mcvc_wgrad_cin2_applies(const ConvProblem& p, const WgradIO& io)
{
    static const int en = mcvc_knob("MCVC_WGRAD_CIN2_MFMA", 1);
    return en && p.Cin >= 1 && p.Cin <= 2 && p.KH == 5 && p.KW == 15 && p.stride == 1 && p.pad_h == 2 && p.pad_w == 7 && p.W == 64 && p.OH == p.H &&
           p.OW == p.W && (io.x_sh & 3) == 0 && (io.x_sc & 3) == 0 && (io.x_sb & 3) == 0 && (io.dy_sh & 3) == 0 && (io.dy_sc & 3) == 0 &&
           (io.dy_sb & 3) == 0 && ((reinterpret_cast<unsigned long long>(io.x) | reinterpret_cast<unsigned long long>(io.dy)) & 15ull) == 0;
}

int mcvc_wgrad_cin2_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, hipStream_t s)
{
    if (!mcvc_wgrad_cin2_applies(p, io)) return MCVC_ERR_INVALID;
    W2Args w{};
    w.x = io.x; w.x_sn = io.x_sb; w.x_sc = io.x_sc; w.x_sh = io.x_sh;
    w.dy = io.dy; w.dy_sn = io.dy_sb; w.dy_sc = io.dy_sc; w.dy_sh = io.dy_sh; w.dw = dw;
    // bands of 4 rows; of 2 rows when that is needed to have a workgroup per CU (one or two samples per pass)
    const int blocks = cdiv_i(p.Cout, kW2Co);
    const int band = ((long long)blocks * NB * cdiv_i(p.H, 4) < 256) ? 2 : 4;
    w.N = NB; w.Cin = p.Cin; w.Cout = p.Cout; w.H = p.H; w.W = p.W; w.bands = cdiv_i(p.H, band); w.units = NB * w.bands;
    int nch = 1;
    // (every workgroup ends with 64 x Cin x 75 atomic adds into the same dW block: enough workgroups to fill the chip, not more)
    static const int wgs = mcvc_knob("MCVC_WGRAD_CIN2_WGS", 512);
    if (!mcvc_deterministic())
        while (2 * nch <= w.units && blocks * nch < wgs) nch *= 2;
    w.atomic = nch > 1 ? 1 : 0;
    size_t lds = (size_t)(kW2Co * (band * 64 + 4) + 2 * (band + 4) * kW2XP) * sizeof(float);
    const size_t lds_out = (size_t)kW2Co * p.Cin * 75 * sizeof(float);             // the write-out tile reuses the staging area
    if (lds < lds_out) lds = lds_out;
    TraceScope ts(K_WGRAD_4x1, s, 2.0 * NB * p.H * p.W * p.Cout * p.Cin * p.KH * p.KW, 4.0 * ((double)NB * p.Cout * p.H * p.W + (double)NB * p.Cin * p.H * p.W));
    const dim3 grid((unsigned)blocks, (unsigned)nch);
    if (p.Cin == 1) { if (band == 2) mcvc_launch((wgrad_cin2_mfma_kernel<1, 2>), grid, dim3(256), lds, s, w); else mcvc_launch((wgrad_cin2_mfma_kernel<1, 4>), grid, dim3(256), lds, s, w); }
    else { if (band == 2) mcvc_launch((wgrad_cin2_mfma_kernel<2, 2>), grid, dim3(256), lds, s, w); else mcvc_launch((wgrad_cin2_mfma_kernel<2, 4>), grid, dim3(256), lds, s, w); }
    return (int)hipGetLastError();
}

bool mcvc_wgrad_cout1_applies(const ConvProblem& p)
{
    if (p.Cout != 1 || p.stride != 1 || p.KH * p.KW > 128 || p.OH != p.H || p.OW != p.W) return false;
    const int XW = p.W + p.KW - 1;
    const size_t lds = ((size_t)(p.H + p.KH - 1) * XW + (size_t)p.H * p.W + 256) * sizeof(float);
    return lds <= 64 * 1024;
}

int mcvc_wgrad_cout1_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, hipStream_t s)
{
    if (!mcvc_wgrad_cout1_applies(p)) return MCVC_ERR_INVALID;
    FewWgradArgs a{};
    a.x = io.x; a.x_sn = io.x_sb; a.x_sc = io.x_sc; a.x_sh = io.x_sh;
    a.dy = io.dy; a.dy_sn = io.dy_sb; a.dy_sh = io.dy_sh;
    a.dw = dw; a.N = NB; a.Cin = p.Cin; a.H = p.H; a.W = p.W; a.KH = p.KH; a.KW = p.KW; a.ph = p.pad_h; a.pw = p.pad_w;
    a.XW = p.W + p.KW - 1;
    // (sample, band of rows) units over gridDim.y workgroups that add their partial sums atomically; deterministic mode: one workgroup per
    // input channel walks whole planes in a fixed order
    // 5 x 15 kernels on 16-byte aligned images: the matrix-core form (MCVC_WGRAD1_MFMA=0: the VALU kernel)
    static const int mfma = mcvc_knob("MCVC_WGRAD1_MFMA", 1);
    if (mfma && p.KH == 5 && p.KW == 15 && p.pad_h == 2 && p.pad_w == 7 && p.W == 64 && (io.x_sh & 3) == 0 && (io.x_sc & 3) == 0 && (io.x_sb & 3) == 0 &&
        (io.dy_sh & 3) == 0 && (io.dy_sb & 3) == 0 && ((reinterpret_cast<unsigned long long>(io.x) | reinterpret_cast<unsigned long long>(io.dy)) & 15ull) == 0) {
        W1Args w{};
        w.x = io.x; w.x_sn = io.x_sb; w.x_sc = io.x_sc; w.x_sh = io.x_sh; w.dy = io.dy; w.dy_sn = io.dy_sb; w.dy_sh = io.dy_sh; w.dw = dw;
        const int tiles = cdiv_i(p.Cin, kW1Ch);
        const int band = ((long long)tiles * NB * cdiv_i(p.H, 8) < 256) ? 4 : 8;
        w.N = NB; w.Cin = p.Cin; w.H = p.H; w.W = p.W; w.bands = cdiv_i(p.H, band); w.units = NB * w.bands;
        int nch = 1;
        if (!mcvc_deterministic())
            while (2 * nch <= w.units && tiles * nch < 1024) nch *= 2;
        w.atomic = nch > 1 ? 1 : 0;
        const size_t lds1 = (size_t)(kW1Ch * ((band + 4) * kW1PW + 4) + band * kW1DyW) * sizeof(float);        // >= the 4 x 5 x 16 x 16 reduction tile
        TraceScope ts(K_WGRAD_SMALLK, s, 2.0 * NB * p.H * p.W * p.Cin * p.KH * p.KW, 4.0 * ((double)NB * p.Cin * p.H * p.W + (double)NB * p.H * p.W * p.Cin));
        if (band == 4) mcvc_launch(wgrad_cout1_mfma_kernel<4>, dim3((unsigned)tiles, (unsigned)nch), dim3(256), lds1, s, w);
        else mcvc_launch(wgrad_cout1_mfma_kernel<8>, dim3((unsigned)tiles, (unsigned)nch), dim3(256), lds1, s, w);
        return (int)hipGetLastError();
    }
    int nchunk = 1;
    a.band = p.H;
    if (!mcvc_deterministic()) {
        while (a.band > 20 && p.Cin * NB * cdiv_i(p.H, a.band) < 512) a.band = (a.band + 1) / 2;
        const int units = NB * cdiv_i(p.H, a.band);
        while (2 * nchunk <= units && nchunk < 32 && p.Cin * nchunk < 768) nchunk *= 2;
    }
    const size_t lds = ((size_t)(a.band + p.KH - 1) * a.XW + (size_t)a.band * p.W + 256) * sizeof(float);
    TraceScope ts(K_WGRAD_SMALLK, s, 2.0 * NB * p.H * p.W * p.Cin * p.KH * p.KW, 4.0 * ((double)NB * p.Cin * p.H * p.W + (double)NB * p.H * p.W * p.Cin));
    mcvc_launch(wgrad_cout1_kernel, dim3((unsigned)p.Cin, (unsigned)nchunk), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

bool mcvc_wgrad_cin1_applies(const ConvProblem& p)
{
    return p.Cin == 1 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1 && p.OH == p.H && p.OW == p.W && p.W <= 512;
}

int mcvc_wgrad_cin1_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, hipStream_t s)
{
    if (!mcvc_wgrad_cin1_applies(p)) return MCVC_ERR_INVALID;
    const int units = NB * cdiv_i(p.H, kCin1Band);
    int nchunk = 1;
    if (!mcvc_deterministic())
        while (2 * nchunk <= units && p.Cout * nchunk < 1024) nchunk *= 2;
    const size_t lds = ((size_t)(kCin1Band + 2) * (p.W + 2) + 36) * sizeof(float);
    TraceScope ts(K_WGRAD_SMALLK, s, 2.0 * NB * p.H * p.W * p.Cout * 9, 4.0 * ((double)NB * p.Cout * p.H * p.W + (double)NB * p.H * p.W));
    mcvc_launch(wgrad_cin1_kernel, dim3((unsigned)p.Cout, (unsigned)nchunk), dim3(256), lds, s, WgradCin1KArgs{io.x, io.x_sb, io.x_sh, io.dy, io.dy_sb, io.dy_sc, io.dy_sh, dw, NB, p.H, p.W});
    return (int)hipGetLastError();
}
