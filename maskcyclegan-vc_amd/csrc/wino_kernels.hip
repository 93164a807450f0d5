// Winograd F(2x2, 5x5) for the generator's stride-1 5x5 convolutions (upSample1: 256->1024, upSample2: 256->512;
// model.py:192-204) -- 64 % of a generator forward's FLOPs.  Interpolation points {0, +-1, +-2, inf}: a 6x6 input tile
// yields a 2x2 output tile with 36 multiplies instead of 100 (2.78x fewer MFMA FLOPs); measured fp32 error of the scheme
// on this layer shape is 1.3e-6 relative (direct fp32 summation: 1.3e-7), far inside the 1e-3 parity bar.
//   V  = B^T d B        (input transform, integers only)        -> [36][Cin][tiles]
//   M_xi = U_xi V_xi    36 independent [Cout x Cin] x [Cin x tiles] products = ONE launch of conv_direct_kernel as a
//                       1x1 convolution over 36 "images" with per-image weights (ConvArgs::w_nstride)
//   Y  = A^T M A + bias (output transform, fused PixelShuffle store)
//   U  = G g G^T        (weight transform, done by the re-pack after each optimizer step)
// The data-gradient of these layers is the same pipeline on flipped, transposed weights.
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include "wino.h"

namespace {

// B^T rows for points {0, 1, -1, 2, -2, inf}
__device__ __forceinline__ void bt6(const float d[6], float o[6])
{
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// one thread = one (channel, tile): 36 loads (rows of 6 consecutive floats), 72 small dot products, 36 coalesced stores
__global__ void __launch_bounds__(256) wino_input_kernel(const Twin<WinoXformArgs> tw)
{
    const WinoXformArgs a = tw.v[blockIdx.z];
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (tile >= a.NT) return;
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    const int ih0 = 2 * ty - a.pad, iw0 = 2 * tx - a.pad;
    const float* src = a.x + (long long)n * a.x_sb + (long long)c * a.x_sc;
    float t[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float d[6];
        const int ih = ih0 + i;
        const bool rok = (ih >= 0) && (ih < a.H);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int iw = iw0 + j;
            d[j] = (rok && iw >= 0 && iw < a.W) ? src[(long long)ih * a.x_sh + iw] : 0.f;
        }
        bt6(d, t[i]);                      // t[i][b] = sum_j d[i][j] BT[b][j]   (transform along the row)
    }
    float* dst = a.v + (long long)c * a.NTp + tile;
    const long long xs = (long long)a.C * a.NTp;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float col[6], o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = t[i][b];
        bt6(col, o);                       // V[aa][b] = sum_i BT[aa][i] t[i][b]
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) dst[(long long)(aa * 6 + b) * xs] = o[aa];
    }
}

// ---- F(2x2,3x3): 4x4 tiles, points {0, 1, -1, inf} -------------------------------------------------------------------
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]   A^T = [[1,1,1,0],[0,1,-1,-1]]
__device__ __forceinline__ void bt4(const float d[4], float o[4])
{
    o[0] = d[0] - d[2]; o[1] = d[1] + d[2]; o[2] = d[2] - d[1]; o[3] = d[1] - d[3];
}

__global__ void __launch_bounds__(256) wino3_input_kernel(const Twin<WinoXformArgs> tw)
{
    const WinoXformArgs a = tw.v[blockIdx.z];
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (tile >= a.NT) return;
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    const int ih0 = 2 * ty - a.pad, iw0 = 2 * tx - a.pad;
    const float* src = a.x + (long long)n * a.x_sb + (long long)c * a.x_sc;
    float t[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float d[4];
        const int ih = ih0 + i;
        const bool rok = (ih >= 0) && (ih < a.H);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = iw0 + j;
            d[j] = (rok && iw >= 0 && iw < a.W) ? src[(long long)ih * a.x_sh + iw] : 0.f;
        }
        bt4(d, t[i]);
    }
    float* dst = a.v + (long long)c * a.NTp + tile;
    const long long xs = (long long)a.C * a.NTp;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float col[4], o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i] = t[i][b];
        bt4(col, o);
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) dst[(long long)(aa * 4 + b) * xs] = o[aa];
    }
}

// phase-plane variant: channel k = 4ci + 2p + q reads x[ci][2i+p][2j+q] of an XH x XW image
struct Wino3InputPhaseKArgs { WinoXformArgs a; int XH; int XW; };
__global__ void __launch_bounds__(256) wino3_input_phase_kernel(const Twin<Wino3InputPhaseKArgs> tw)
{
    const Wino3InputPhaseKArgs ka_ = tw.v[blockIdx.z];
    const WinoXformArgs& a = ka_.a;
    int XH = ka_.XH;
    int XW = ka_.XW;
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (tile >= a.NT) return;
    const int ci = k >> 2, p = (k >> 1) & 1, q = k & 1;
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    const int i0 = 2 * ty - 1, j0 = 2 * tx - 1;                 // phase-plane coordinates (padding 1)
    const float* src = a.x + (long long)n * a.x_sb + (long long)ci * a.x_sc;
    float t[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float d[4];
        const int ih = 2 * (i0 + i) + p;
        const bool rok = (i0 + i >= 0) && (ih < XH);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = 2 * (j0 + j) + q;
            d[j] = (rok && j0 + j >= 0 && iw < XW) ? src[(long long)ih * a.x_sh + iw] : 0.f;
        }
        bt4(d, t[i]);
    }
    float* dst = a.v + (long long)k * a.NTp + tile;
    const long long xs = (long long)a.C * a.NTp;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float col[4], o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i] = t[i][b];
        bt4(col, o);
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) dst[(long long)(aa * 4 + b) * xs] = o[aa];
    }
}

// tile-major twins for the weight gradient (16 tiles x 16 channels per workgroup, channel fastest)

// A3 (4x2) = [[1,0],[1,1],[1,-1],[0,-1]]:  dM = A3 dy A3^T
__device__ __forceinline__ void a42(float d0, float d1, float o[4]) { o[0] = d0; o[1] = d0 + d1; o[2] = d0 - d1; o[3] = -d1; }


// dg' = G3^T dU G3 (3x3 per (co, k = 4ci+2p+q)), scattered into the OIHW gradient: dw[co][ci][2u'+p][2v'+q] += dg'[u'][v']
struct Wino3DwKArgs { const float* du; float* dw0; float* dw1; int Cout; int nbr; int Cin; };
__global__ void __launch_bounds__(256) wino3_dw_kernel(const Twin<Wino3DwKArgs> tw)
{
    const Wino3DwKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ du = ka_.du;
    float* __restrict__ dw0 = ka_.dw0;
    float* __restrict__ dw1 = ka_.dw1;
    int Cout = ka_.Cout;
    int nbr = ka_.nbr;
    int Cin = ka_.Cin;
    const int k = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    const int K = 4 * Cin;
    if (k >= K) return;
    const long long xs = (long long)Cout * nbr * K;
    const float* src = du + (long long)co * K + k;
    auto gt3 = [](const float v[4], float o[3]) {
        o[0] = v[0] + 0.5f * (v[1] + v[2]); o[1] = 0.5f * (v[1] - v[2]); o[2] = 0.5f * (v[1] + v[2]) + v[3];
    };
    float t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float col[4], o[3];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) col[aa] = src[(long long)(aa * 4 + b) * xs];
        gt3(col, o);
#pragma unroll
        for (int u = 0; u < 3; ++u) t[u][b] = o[u];
    }
    const int ci = k >> 2, p = (k >> 1) & 1, q = k & 1;
    float* dw = (co < Cout) ? dw0 : dw1;
    const int col = (co < Cout) ? co : co - Cout;
    if (!dw) return;
    float* dst = dw + ((long long)col * Cin + ci) * 25;
    // (the nine read-modify-writes: all loads before the first store -- written as `dst[..] += o` they run as nine sequential round trips)
    float r[3][3], old[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u) gt3(t[u], r[u]);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int kh = 2 * u + p, kw = 2 * v + q;
            old[u][v] = (kh <= 4 && kw <= 4) ? dst[kh * 5 + kw] : 0.f;
        }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int kh = 2 * u + p, kw = 2 * v + q;
            if (kh <= 4 && kw <= 4) dst[kh * 5 + kw] = old[u][v] + r[u][v];
        }
}

__global__ void __launch_bounds__(256) wino3_output_kernel(const Twin<WinoOutArgs> tw)
{
    const WinoOutArgs a = tw.v[blockIdx.z];
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int co = blockIdx.y;
    if (tile >= a.NT) return;
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    const float* src = a.m + (long long)co * a.NTp + tile;
    const long long xs = (long long)a.Cout * a.NTp;
    float u[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float m[4];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) m[aa] = src[(long long)(aa * 4 + b) * xs];
        u[0][b] = m[0] + m[1] + m[2];
        u[1][b] = m[1] - m[2] - m[3];
    }
    const float bias = a.bias ? a.bias[co] : 0.f;
    float* yn = a.y + (long long)n * a.y_sb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float ov[2] = {u[i][0] + u[i][1] + u[i][2] + bias, u[i][1] - u[i][2] - u[i][3] + bias};
        const int oh = 2 * ty + i;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ow = 2 * tx + j;
            if (oh >= a.OH || ow >= a.OW) continue;
            long long off;
            if (a.shuffle) {
                const int yh = 2 * oh + ((co >> 1) & 1), yw = 2 * ow + (co & 1);
                if (yh >= a.YH || yw >= a.YW) continue;
                off = (long long)(co >> 2) * a.y_sc + (long long)yh * a.y_sh + yw;
            } else {
                off = (long long)co * a.y_sc + (long long)oh * a.y_sh + ow;
            }
            if (a.accumulate) yn[off] += ov[j]; else yn[off] = ov[j];
        }
    }
}

// ---- weight-gradient operands --------------------------------------------------------------------------------------
// dU[xi][co][ci] = sum_tile dM[xi][co][tile] * V[xi][ci][tile]: the tile index is the contraction dimension, so both
// operands are stored tile-major ([xi][tile][channel]) -- the K-major layout the batched GEMM streams.  A workgroup
// covers 16 tiles x 16 channels with the channel fastest, so every store is a 64-byte run.

// A (6x2) = [[1,0],[1,1],[1,-1],[1,2],[1,-2],[0,1]]:  dM = A dy A^T
__device__ __forceinline__ void a62(float d0, float d1, float o[6])
{
    o[0] = d0; o[1] = d0 + d1; o[2] = d0 - d1; o[3] = d0 + 2.f * d1; o[4] = d0 - 2.f * d1; o[5] = d1;
}


// ---- tile-major operands through an LDS transpose (r3) ---------------------------------------------------------------------------
// The kernels above give a lane one (tile, channel) and let 16 lanes walk the channels: 64-byte stores and reads scattered over 16
// channel planes (1.7 TB/s at bs=1, 2-3 TB/s at bs=32 for outputs 9x / 4x the size of their input).  Here a workgroup owns 8 consecutive
// tiles x 64 channels: lanes walk the TILES while reading (a channel's 8 tiles are one 64-byte run of a row), the transformed values
// go through LDS [point][tile][channel], and every store is a 256-byte run of 64 channels.
//   KIND 0: V^T of the F(2x2,5x5) input (6x6 window, pad)      1: dM^T = A dY A^T, 36 points
//   KIND 2: V^T of the phase-plane F(2x2,3x3) input            3: dM^T, 16 points
struct XformTKArgs { WinoXformArgs a; int XH; int XW; };
constexpr int kXT = 8, kXC = 64, kXPitch = 68;

template <int KIND>
__device__ __forceinline__ void xform_t_compute(const WinoXformArgs& a, int XH, int XW, int tile, int c, float* o)
{
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    if constexpr (KIND == 0) {
        const int ih0 = 2 * ty - a.pad, iw0 = 2 * tx - a.pad;
        const float* src = a.x + (long long)n * a.x_sb + (long long)c * a.x_sc;
        float t[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float d[6];
            const int ih = ih0 + i;
            const bool rok = (ih >= 0) && (ih < a.H);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int iw = iw0 + j;
                d[j] = (rok && iw >= 0 && iw < a.W) ? src[(long long)ih * a.x_sh + iw] : 0.f;
            }
            bt6(d, t[i]);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float col[6], q[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = t[i][b];
            bt6(col, q);
#pragma unroll
            for (int aa = 0; aa < 6; ++aa) o[aa * 6 + b] = q[aa];
        }
    } else if constexpr (KIND == 2) {
        const int ci = c >> 2, p = (c >> 1) & 1, q = c & 1;
        const int i0 = 2 * ty - 1, j0 = 2 * tx - 1;
        const float* src = a.x + (long long)n * a.x_sb + (long long)ci * a.x_sc;
        float t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float d[4];
            const int ih = 2 * (i0 + i) + p;
            const bool rok = (i0 + i >= 0) && (ih < XH);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = 2 * (j0 + j) + q;
                d[j] = (rok && j0 + j >= 0 && iw < XW) ? src[(long long)ih * a.x_sh + iw] : 0.f;
            }
            bt4(d, t[i]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float col[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) col[i] = t[i][b];
            bt4(col, w);
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) o[aa * 4 + b] = w[aa];
        }
    } else {
        const float* src = a.x + (long long)n * a.x_sb + (long long)c * a.x_sc;
        float dy[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oh = 2 * ty + i, ow = 2 * tx + j;
                dy[i][j] = (oh < a.H && ow < a.W) ? src[(long long)oh * a.x_sh + ow] : 0.f;
            }
        if constexpr (KIND == 1) {
            float t0[6], t1[6];
            a62(dy[0][0], dy[1][0], t0);
            a62(dy[0][1], dy[1][1], t1);
#pragma unroll
            for (int aa = 0; aa < 6; ++aa) a62(t0[aa], t1[aa], o + aa * 6);
        } else {
            float t0[4], t1[4];
            a42(dy[0][0], dy[1][0], t0);
            a42(dy[0][1], dy[1][1], t1);
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) a42(t0[aa], t1[aa], o + aa * 4);
        }
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) xform_t_kernel(const Twin<XformTKArgs> tw)
{
    const XformTKArgs ka_ = tw.v[blockIdx.z];
    const WinoXformArgs& a = ka_.a;
    constexpr int P = (KIND < 2) ? 36 : 16;
    extern __shared__ __attribute__((aligned(16))) float xbuf[];          // [P][kXT][kXPitch]
    const int tid = threadIdx.x;
    const int tile0 = blockIdx.x * kXT, c0 = blockIdx.y * kXC;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int tl = tid & 7, cl = (tid >> 3) + 32 * it;
        const int tile = tile0 + tl, c = c0 + cl;
        float o[P];
        if (tile < a.NT && c < a.C) xform_t_compute<KIND>(a, ka_.XH, ka_.XW, tile, c, o);
        else {
#pragma unroll
            for (int q = 0; q < P; ++q) o[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < P; ++q) xbuf[(q * kXT + tl) * kXPitch + cl] = o[q];
    }
    __syncthreads();
    const int cl = tid & 63, c = c0 + cl;
    if (c < a.C) {
#pragma unroll 4
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tl = (tid >> 6) + 4 * h, tile = tile0 + tl;
                if (tile < a.NTp) a.v[((long long)q * a.NTp + tile) * a.C + c] = xbuf[(q * kXT + tl) * kXPitch + cl];
            }
    }
}

template <int KIND>
static int xform_t_launch(const WinoXformArgs& a, int XH, int XW, double bytes, hipStream_t s)
{
    constexpr int P = (KIND < 2) ? 36 : 16;
    constexpr size_t lds = (size_t)P * kXT * kXPitch * sizeof(float);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xform_t_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    dim3 grid((unsigned)cdiv_i(a.NTp, kXT), (unsigned)cdiv_i(a.C, kXC));
    TraceScope ts(K_ELEMENTWISE, s, 0.0, bytes);
    mcvc_launch(xform_t_kernel<KIND>, grid, dim3(256), lds, s, XformTKArgs{a, XH, XW});
    return (int)hipGetLastError();
}

// dg = G^T dU G, accumulated into the OIHW gradient.  One thread per (co, ci), ci fastest (coalesced reads of dU).
struct WinoDwKArgs { const float* du; float* dw; int Cout; int Cin; };
__global__ void __launch_bounds__(256) wino_dw_kernel(const Twin<WinoDwKArgs> tw)
{
    const WinoDwKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ du = ka_.du;
    float* __restrict__ dw = ka_.dw;
    int Cout = ka_.Cout;
    int Cin = ka_.Cin;
    // (the 25 results of a thread go through LDS so that the read-modify-write of dW -- 256 input channels x 25 taps of one output channel
    // are 6400 CONSECUTIVE floats -- runs over consecutive addresses instead of 25 accesses at a 100-byte stride per thread)
    __shared__ float wt[256 * 25];
    const int ci = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    const bool live = ci < Cin;
    const long long xs = (long long)Cout * Cin;
    const float* src = du + (long long)co * Cin + (live ? ci : 0);
    // G^T (5x6) applied to a 6-vector:  out[k] = sum_a G[a][k] v[a]
    auto gt = [](const float v[6], float o[5]) {
        const float p = v[1] + v[2], m = v[1] - v[2], q = v[3] + v[4], n = v[3] - v[4];
        o[0] = 0.25f * v[0] - p * (1.0f / 6.0f) + q * (1.0f / 24.0f);
        o[1] = -m * (1.0f / 6.0f) + n * (1.0f / 12.0f);
        o[2] = -p * (1.0f / 6.0f) + q * (1.0f / 6.0f);
        o[3] = -m * (1.0f / 6.0f) + n * (1.0f / 3.0f);
        o[4] = -p * (1.0f / 6.0f) + q * (2.0f / 3.0f) + v[5];
    };
    float t[5][6];                            // t[k][b] = sum_a G[a][k] dU[a][b]
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float col[6], o[5];
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) col[aa] = src[(long long)(aa * 6 + b) * xs];
        gt(col, o);
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k][b] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float o[5];
        gt(t[k], o);                          // dg[k][l] = sum_b t[k][b] G[b][l]
#pragma unroll
        for (int l = 0; l < 5; ++l) wt[threadIdx.x * 25 + k * 5 + l] = o[l];
    }
    __syncthreads();
    int nci = Cin - (int)blockIdx.x * 256; if (nci > 256) nci = 256;
    float* dst = dw + ((long long)co * Cin + (long long)blockIdx.x * 256) * 25;
    float old[25];                               // (all loads of the read-modify-write before the first store)
#pragma unroll
    for (int u = 0; u < 25; ++u) { const int i = threadIdx.x + u * 256; old[u] = (i < nci * 25) ? dst[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 25; ++u) { const int i = threadIdx.x + u * 256; if (i < nci * 25) dst[i] = old[u] + wt[i]; }
}

// A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,1]]
__device__ __forceinline__ void at6(const float m[6], float& o0, float& o1)
{
    o0 = m[0] + m[1] + m[2] + m[3] + m[4];
    o1 = m[1] - m[2] + 2.f * (m[3] - m[4]) + m[5];
}

__global__ void __launch_bounds__(256) wino_output_kernel(const Twin<WinoOutArgs> tw)
{
    const WinoOutArgs a = tw.v[blockIdx.z];
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int co = blockIdx.y;
    if (tile >= a.NT) return;
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    const float* src = a.m + (long long)co * a.NTp + tile;
    const long long xs = (long long)a.Cout * a.NTp;
    float u[2][6];                                  // u[i][b] = sum_a AT[i][a] M[a][b]
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float col[6];
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) col[aa] = src[(long long)(aa * 6 + b) * xs];
        at6(col, u[0][b], u[1][b]);
    }
    const float bias = a.bias ? a.bias[co] : 0.f;
    float* yn = a.y + (long long)n * a.y_sb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float o0, o1;
        at6(u[i], o0, o1);
        const float ov[2] = {o0 + bias, o1 + bias};
        const int oh = 2 * ty + i;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ow = 2 * tx + j;
            if (oh >= a.OH || ow >= a.OW) continue;
            long long off;
            if (a.shuffle) {
                const int yh = 2 * oh + ((co >> 1) & 1), yw = 2 * ow + (co & 1);
                if (yh >= a.YH || yw >= a.YW) continue;
                off = (long long)(co >> 2) * a.y_sc + (long long)yh * a.y_sh + yw;
            } else {
                off = (long long)co * a.y_sc + (long long)oh * a.y_sh + ow;
            }
            if (a.accumulate) yn[off] += ov[j]; else yn[off] = ov[j];
        }
    }
}


// ---- batched product of the 36 points -----------------------------------------------------------------------------
// Workgroup = 4 waves, tile 128 (m) x 64 (n); wave (wm, wn) owns 64 x 32 = two v_mfma_f32_32x32x2_f32 accumulators.
// K advances in stages of 16; both operands are plain K-major matrices, so a stage is two rectangular copies
// (16 x 128 and 16 x 64 floats) done by LDS-DMA, 16 bytes per lane.  FOUR stages are in flight: unlike the conv kernel
// (one stage of lookahead, drained by the vmcnt(0) that __syncthreads() implies) the wait here is an explicit
// s_waitcnt vmcnt(n) that leaves the newer stages outstanding, so a stage has three stages of MFMA work (~2.5 us) to land.
constexpr int kGK = 16;                 // k per stage
constexpr int kGStages = 4;
#ifndef MCVC_FOLDK
#define MCVC_FOLDK 32
#endif
constexpr int kFoldK = MCVC_FOLDK;              // k per first-level accumulation chain of the batched GEMMs (two-level sums: wino_gemm_kernel)
constexpr int kGA = kGK * 128;          // floats of A per stage

__device__ __forceinline__ void wg_glds16(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// BN = 64: waves 2 (m) x 2 (n), two accumulators each (64 x 32).  BN = 32 (r2): waves 4 (m) x 1, one accumulator each (32 x 32) -- for
// the ragged tile counts of upSample1 / downSample2 at B <= 2 (N = 96, 160 columns): with 64-column tiles a quarter of the MFMA work
// was padding and 576 workgroups of 6.8 us left a third of the chip idle in the last round; 32-column tiles are exact, half the size,
// and four of them fit a CU (40 KB of LDS each).
template <int BN>
__global__ void __launch_bounds__(256, (BN == 64 ? 3 : 4)) wino_gemm_kernel(const Twin<WinoGemmArgs> tw)
{
    const WinoGemmArgs a = tw.v[blockIdx.z];
    constexpr int kGB = kGK * BN;
    constexpr int kGStage = kGA + kGB;
    constexpr int NACC = (BN == 64) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [kGStages][A 16x128 | B 16xBN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = (BN == 64) ? (wave >> 1) : wave, wn = (BN == 64) ? (wave & 1) : 0;
    // XCD-aware order (8 XCDs, private L2 each; hardware deals consecutive workgroup ids round-robin to the XCDs): every XCD
    // gets one contiguous range of logical ids, and logical ids run n-tile fastest, then m-tile, then point -- so the
    // workgroups that share an A panel (same m, xi) or a B panel (same n, xi) sit behind the same L2.  PMC before this
    // remap: 132 MB fetched per launch against 54 MB of operands.
    int lid;
    {
        const int total = (int)gridDim.x, linear = (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = linear & 7, k = linear >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n0 = (lid % a.nt) * BN; lid /= a.nt;
    const int m0 = (lid % a.mt) * 128;
    const int xi = lid / a.mt;
    const float* A = a.a + (long long)xi * a.a_xi + m0;
    const float* B = a.b + (long long)xi * a.b_xi + n0;
    // DMA lane constants.  A stage: 16 rows x 32 float4 = 512 float4 = 8 wave-instructions (2 per wave);
    //                      B stage: 16 rows x BN/4 float4: BN = 64 -> 4 wave-instructions (1 per wave); BN = 32 -> 2, issued by
    //                      waves 0-1 and DUPLICATED by waves 2-3 (same bytes to the same LDS words) so that every wave has exactly
    //                      three DMA instructions per stage in flight and one vmcnt immediate serves all of them
    const int fa0 = wave * 64 + lane, fa1 = fa0 + 256;                // float4 index inside the A stage
    const float* a_src0 = A + (long long)(fa0 >> 5) * a.lda + 4 * (fa0 & 31);
    const float* a_src1 = A + (long long)(fa1 >> 5) * a.lda + 4 * (fa1 & 31);
    const int bw = (BN == 64) ? wave : (wave & 1);
    const int fb = bw * 64 + lane;
    constexpr int F4R = BN / 4;                                       // float4 per B row
    int bcol = 4 * (fb % F4R);
    if (n0 + bcol > a.ldb - 4) bcol = a.ldb - 4 - n0;                 // tile hanging over the row end: finite neighbours
    const float* b_src = B + (long long)(fb / F4R) * a.ldb + bcol;
    const long long a_step = (long long)kGK * a.lda, b_step = (long long)kGK * a.ldb;
    auto issue = [&](int stage_k, int buf) {                          // 3 DMA instructions per wave
        float* base = smem + buf * kGStage;
        wg_glds16(a_src0 + stage_k * a_step, base + (wave * 64) * 4);
        wg_glds16(a_src1 + stage_k * a_step, base + (wave * 64 + 256) * 4);
        wg_glds16(b_src + stage_k * b_step, base + kGA + (bw * 64) * 4);
    };
    // Two-level accumulation (r6): `acc` runs over kFoldK k, then folds into `tot`.  An fp32 chain of L fused multiply-adds carries a
    // rounding error ~ eps * L / sqrt(2) of one term; in the Winograd domain that error is amplified by the cancellation of the output
    // transform, and the K-long chain was most of the schemes' error (tools/parity_probe.py ops: profiles/r06_parity_ops_{before,after}_fold.log).  Chains of 32 + a chain of c = K / 32 partial
    // sums: error ~ sqrt(L^2 / c + L c) instead of L: 2.5x smaller at K = 256, 3.3x at 512, 4x at 1024 (the optimum is c = sqrt(L)).
    f32x16 acc[NACC], tot[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][r] = 0.f; }
    constexpr int kFoldSt = kFoldK / kGK;
    const int nst = a.K / kGK;
#pragma unroll
    for (int s = 0; s < kGStages - 1; ++s)
        if (s < nst) issue(s, s);
    const int a_lane = half * 128 + wm * (32 * NACC) + l31;           // A[k = 2p + half][m]
    const int b_lane = kGA + half * BN + wn * 32 + l31;               // B[k = 2p + half][n]
    for (int st = 0; st < nst; ++st) {
        // stage `st` must have landed: at most the (up to) two newer stages' 3 + 3 DMA instructions may stay outstanding
        const int newer = (nst - 1 - st) < (kGStages - 2) ? (nst - 1 - st) : (kGStages - 2);
        if (newer >= 2) __builtin_amdgcn_s_waitcnt(0x0F76);           // vmcnt(6)
        else if (newer == 1) __builtin_amdgcn_s_waitcnt(0x0F73);      // vmcnt(3)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
        __builtin_amdgcn_s_barrier();                                 // everyone's part of stage st is in LDS; buffer (st-1)%4 is free
        if (st + kGStages - 1 < nst) issue(st + kGStages - 1, (st + kGStages - 1) % kGStages);
        if (st != 0 && (st % kFoldSt) == 0) {                         // (behind the barrier: the chunk's last MFMAs have retired by now)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot[i][r] += acc[i][r]; acc[i][r] = 0.f; }
        }
        const float* sb = smem + (st % kGStages) * kGStage;
        // operand reads run one k-pair ahead of the MFMAs (pinned: the scheduler otherwise sinks them below the MFMAs)
        if constexpr (BN == 64) {
            float a0 = sb[a_lane], a1 = sb[a_lane + 32], b0 = sb[b_lane];
#pragma unroll
            for (int p = 0; p < kGK / 2; ++p) {
                const int q = (p + 1 < kGK / 2) ? p + 1 : p;
                const float na0 = sb[a_lane + q * 256], na1 = sb[a_lane + q * 256 + 32], nb0 = sb[b_lane + q * 2 * BN];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                a0 = na0; a1 = na1; b0 = nb0;
            }
        } else {
            float a0 = sb[a_lane], b0 = sb[b_lane];
#pragma unroll
            for (int p = 0; p < kGK / 2; ++p) {
                const int q = (p + 1 < kGK / 2) ? p + 1 : p;
                const float na0 = sb[a_lane + q * 256], nb0 = sb[b_lane + q * 2 * BN];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                a0 = na0; b0 = nb0;
            }
        }
    }
    // ---- store: D register r of lane (l31, half) is row (r&3) + 8*(r>>2) + 4*half, column l31
    // (M is a multiple of the row tile: no row check.  The row pitch is read ONCE, in front of the stores: with a per-row `m < M` test the
    // compiler sank the kernarg load of ldc into each of the 16 conditional stores -- s_load + s_waitcnt per store, ~1 us of a short-K tile)
    const int n = n0 + wn * 32 + l31;
    const long long ldc = a.ldc;
    if (n < a.N) {
        float* C = a.c + (long long)xi * a.c_xi + n + (long long)(m0 + wm * (32 * NACC) + 4 * half) * ldc;
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) C[(long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = tot[i][r] + acc[i][r];
    }
}

// ---- generalised tile / stage shapes (r2 tuning: tools/gemm_tune.py) ------------------------------------------------------------
// BM x BN output tile, GK k per stage, ST stages in flight.  Waves: BM=128,BN=64 -> 2x2 of 64x32 (two accumulators); BM=128,BN=32 ->
// 4x1 of 32x32; BM=64,BN=64 -> 2x2 of 32x32.  Every wave issues the same number of DMA instructions per stage (fractional shares are
// duplicated), so one vmcnt immediate per "stages still in flight" serves all waves.
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

template <int BM, int BN, int GK, int ST>
__global__ void __launch_bounds__(256) gemm2_kernel(const Twin<WinoGemmArgs> tw)
{
    const WinoGemmArgs a = tw.v[blockIdx.z];
    constexpr int SA = GK * BM, SB = GK * BN, STAGE = SA + SB;
    // waves 2 x 2 when BN is a multiple of 64, else 4 x 1 (every wave 32 rows x all BN columns: the one-tile-wide shapes for N = 96 / 160)
    constexpr bool W22 = (BN % 64) == 0;
    constexpr int NBM = W22 ? BM / 64 : BM / 128, NBN = W22 ? BN / 64 : BN / 32, NACC = NBM * NBN;     // 32x32 blocks per wave (rows x columns)
    constexpr int NA = (SA / 4 + 255) / 256, NB = (SB / 4 + 255) / 256, ND = NA + NB;     // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = W22 ? (wave >> 1) : wave, wn = W22 ? (wave & 1) : 0;
    int lid;
    {
        const int total = (int)gridDim.x, linear = (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = linear & 7, k = linear >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // Tile order inside a point (r6): with `mgroup` row tiles per group the ROW tile is the fastest index inside a group, then the column tile
    // (igemm_kernel's order) -- the ~64 workgroups an XCD holds at a time are then mgroup x (64 / mgroup) tiles that sweep k side by side, so
    // a stage's A chunk is fetched once for 64 / mgroup of them and its B chunk once for mgroup (the reuse is SIMULTANEOUS: it needs no L2
    // capacity), against 1 x 64 tiles -- B fetched once per row tile -- with the column tile fastest (mgroup = 0).
    int n_t, m_t, xi;
    if (a.mgroup == 0) { n_t = lid % a.nt; lid /= a.nt; m_t = lid % a.mt; xi = lid / a.mt; }
    else {
        const int mg = a.mgroup, per = a.mt * a.nt;
        int r = lid % per;
        xi = lid / per;
        const int g = r / (mg * a.nt);
        const int gm = (g + 1) * mg <= a.mt ? mg : a.mt - g * mg;
        r -= g * mg * a.nt;
        n_t = r / gm; m_t = g * mg + r % gm;
    }
    const int n0 = n_t * BN, m0 = m_t * BM;
    const float* A = a.a + (long long)xi * a.a_xi + m0;
    const float* B = a.b + (long long)xi * a.b_xi + n0;
    constexpr int A4R = BM / 4, B4R = BN / 4;                     // float4 per row
    const float* asrc[NA]; int adst[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = (tid + i * 256) % (SA / 4);                 // (duplicates when the stage has fewer than 256 float4 per sweep)
        asrc[i] = A + (long long)(f / A4R) * a.lda + 4 * (f % A4R);
        adst[i] = (((wave * 64 + i * 256) % (SA / 4))) * 4;      // wave-uniform LDS base (float index) of this instruction
    }
    const float* bsrc[NB]; int bdst[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = (tid + i * 256) % (SB / 4);
        int bcol = 4 * (f % B4R);
        if (n0 + bcol > a.ldb - 4) bcol = a.ldb - 4 - n0;
        bsrc[i] = B + (long long)(f / B4R) * a.ldb + bcol;
        bdst[i] = SA + (((wave * 64 + i * 256) % (SB / 4))) * 4;
    }
    const long long a_step = (long long)GK * a.lda, b_step = (long long)GK * a.ldb;
    auto issue = [&](int stage_k, int buf) {
        float* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) wg_glds16(asrc[i] + stage_k * a_step, base + adst[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) wg_glds16(bsrc[i] + stage_k * b_step, base + bdst[i]);
    };
    f32x16 acc[NACC], tot[NACC];                 // two-level accumulation: see wino_gemm_kernel
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][r] = 0.f; }
    constexpr int kFoldSt = kFoldK / GK;
    const int nst = a.K / GK;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nst) issue(s, s);
    const int a_lane = half * BM + wm * (32 * NBM) + l31;
    const int b_lane = SA + half * BN + wn * (32 * NBN) + l31;
    for (int st = 0; st < nst; ++st) {
        const int newer = (nst - 1 - st) < (ST - 2) ? (nst - 1 - st) : (ST - 2);
        if (newer >= 2) wait_vm<2 * ND>(); else if (newer == 1) wait_vm<ND>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + ST - 1 < nst) issue(st + ST - 1, (st + ST - 1) % ST);
        if (st != 0 && (st % kFoldSt) == 0) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot[i][r] += acc[i][r]; acc[i][r] = 0.f; }
        }
        const float* sb = smem + (st % ST) * STAGE;
        float av[NBM], bv[NBN];
#pragma unroll
        for (int i = 0; i < NBM; ++i) av[i] = sb[a_lane + 32 * i];
#pragma unroll
        for (int j = 0; j < NBN; ++j) bv[j] = sb[b_lane + 32 * j];
#pragma unroll
        for (int p = 0; p < GK / 2; ++p) {
            const int q = (p + 1 < GK / 2) ? p + 1 : p;
            float na[NBM], nb[NBN];
#pragma unroll
            for (int i = 0; i < NBM; ++i) na[i] = sb[a_lane + q * 2 * BM + 32 * i];
#pragma unroll
            for (int j = 0; j < NBN; ++j) nb[j] = sb[b_lane + q * 2 * BN + 32 * j];
#pragma unroll
            for (int i = 0; i < NBM; ++i)
#pragma unroll
                for (int j = 0; j < NBN; ++j) acc[i * NBN + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i * NBN + j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NBM + NBN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);
#pragma unroll
            for (int i = 0; i < NBM; ++i) av[i] = na[i];
#pragma unroll
            for (int j = 0; j < NBN; ++j) bv[j] = nb[j];
        }
    }
    const long long ldc = a.ldc;                 // (read once; M % BM == 0: no row check -- see wino_gemm_kernel)
    const int Nv = a.N;
    float* Cb = a.c + (long long)xi * a.c_xi + (long long)(m0 + wm * (32 * NBM) + 4 * half) * ldc;
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
        const int n = n0 + wn * (32 * NBN) + 32 * j + l31;
        if (n >= Nv) continue;
        float* C = Cb + n;
#pragma unroll
        for (int i = 0; i < NBM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) C[(long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = tot[i * NBN + j][r] + acc[i * NBN + j][r];
    }
}

template <int BM, int BN, int GK, int ST>
static int gemm2_launch(WinoGemmArgs b, int nxi, hipStream_t s)
{
    if ((b.M % BM) != 0 || (b.K % GK) != 0) return MCVC_ERR_INVALID;
    b.nt = cdiv_i(b.N, BN); b.mt = b.M / BM;
    {   // row tiles per group (see gemm2_kernel): where a point has enough column tiles for the order to matter.  FETCH_SIZE per launch of the
        // 128 x 128 products at 32 samples 720 -> 356 MB, at 8 samples 203 -> 164 MB (profiles/r06b_pmc_order.log); the one- and two-sample
        // products (1-3 column tiles) keep the column tile fastest
        static const int mgk = mcvc_knob("MCVC_GEMM_MGROUP", 8);
        static const int mgn = mcvc_knob("MCVC_GEMM_MGROUP_MINNT", 8);
        b.mgroup = (mgk > 0 && b.nt >= mgn) ? (mgk < b.mt ? mgk : b.mt) : 0;
    }
    constexpr size_t lds = (size_t)ST * (GK * BM + GK * BN) * sizeof(float);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_kernel<BM, BN, GK, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    mcvc_launch((gemm2_kernel<BM, BN, GK, ST>), dim3((unsigned)(b.nt * b.mt * nxi)), dim3(256), lds, s, b);
    return (int)hipGetLastError();
}

}  // namespace

int mcvc_wino_input_launch(const WinoXformArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.C);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * a.C * a.H * a.W + 36.0 * a.C * a.NT));
    mcvc_launch(wino_input_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_wino_output_launch(const WinoOutArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.Cout);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (36.0 * a.Cout * a.NT + 4.0 * a.Cout * a.NT));
    mcvc_launch(wino_output_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_wino_gemm_launch(const WinoGemmArgs& a, hipStream_t s)
{
    if ((a.K % kGK) != 0 || (a.M % 128) != 0 || (a.lda & 3) || (a.ldb & 3) || a.ldb < 64) return MCVC_ERR_INVALID;
    const int nxi = a.nxi > 0 ? a.nxi : 36;
    WinoGemmArgs b = a;
    // 32-column tiles when 64-column tiles would pad the N range by more than 10 % on a short range (ragged tile counts at B <= 2)
    static const int knob = mcvc_knob("MCVC_WINO_BN", 0);
    const int pad64 = cdiv_i(a.N, 64) * 64, pad32 = cdiv_i(a.N, 32) * 32;
    bool narrow = (pad64 * 10 > pad32 * 11) && a.N <= 512 && a.ldb >= 32;
    if (knob == 32) narrow = a.ldb >= 32; else if (knob == 64) narrow = false;
    b.mt = a.M / 128;
    TraceScope ts(K_WINO_GEMM, s, 2.0 * nxi * a.M * a.N * a.K, 4.0 * nxi * ((double)a.K * a.M + (double)a.K * a.N + (double)a.M * a.N));
#ifdef MCVC_EXPERIMENTS                      // tile sweeps of tools/gemm_*.py: force one gemm2 shape (1-11: r2 sweep; 12-15: 128 x 128 tiles)
    static const int cfg2 = mcvc_knob("MCVC_GEMM_CFG", 0);
    static const bool verbose = mcvc_knob_set("MCVC_GEMM_VERBOSE");        // one line per product: shapes for tools/gemm_bs1_shapes.py
    if (verbose && mcvc_twin_phase() != 1) fprintf(stderr, "[gemm] nxi=%d M=%d N=%d K=%d lda=%d ldb=%d twin=%d\n", nxi, a.M, a.N, a.K, a.lda, a.ldb, mcvc_twin_phase() == 2);
    switch (cfg2) {
        case 1: return gemm2_launch<128, 64, 16, 4>(b, nxi, s);
        case 2: return gemm2_launch<128, 32, 16, 4>(b, nxi, s);
        case 3: return gemm2_launch<64, 64, 16, 4>(b, nxi, s);
        case 4: return gemm2_launch<128, 64, 32, 3>(b, nxi, s);
        case 5: return gemm2_launch<128, 32, 32, 3>(b, nxi, s);
        case 6: return gemm2_launch<64, 64, 32, 3>(b, nxi, s);
        case 7: return gemm2_launch<128, 64, 16, 6>(b, nxi, s);
        case 8: return gemm2_launch<128, 32, 16, 6>(b, nxi, s);
        case 9: return gemm2_launch<64, 64, 16, 6>(b, nxi, s);
        case 10: return gemm2_launch<128, 64, 32, 4>(b, nxi, s);
        case 11: return gemm2_launch<64, 64, 32, 4>(b, nxi, s);
        case 12: return gemm2_launch<128, 128, 16, 4>(b, nxi, s);
        case 13: return gemm2_launch<128, 128, 32, 3>(b, nxi, s);
        case 14: return gemm2_launch<128, 128, 16, 6>(b, nxi, s);
        case 15: return gemm2_launch<128, 128, 32, 2>(b, nxi, s);
        default: break;
    }
#else
    constexpr int cfg2 = 0;
#endif
    // long-K products (the F(2x2,3x3) layers: K = 4*Cin or Cout >= 512): 64x64 tiles with 32-deep stages measured 10-20 % faster than
    // 128x64 / 128x32 with 16-deep stages (profiles/r02_gemm_tune.log) -- twice the workgroups and half the barriers per k
    // ... unless the narrow tiles leave the busiest compute unit clearly less to do (r4, profiles/r04b_gemm_bs1_shapes.log): every product of
    // these sizes follows  time ~ ceil(workgroups / 256) x BM x BN x K  -- the tile-rounds of the fullest CU.  The one- and two-sample
    // data-gradient products (N = 96 / 160 columns, M = 256 / 1024) take 4 rounds of 64x64 tiles (N padded to 128 / 192) against 3 of 128x32:
    // 44 -> 36, 61 -> 45, 44 -> 36 us at one sample, 58 -> 52, 73 -> 59, 58 -> 52 us at two.
    // large grids (from ~16 samples per pass): 128 x 128 tiles, 2 x 2 accumulators per wave -- half the L2 -> LDS bytes per FLOP of the 64 x 64 /
    // 128 x 64 tiles and one operand read per MFMA instead of 1.5-2.  r4, cold operands, alone on the chip: 36 x 512 x 2560 x 512 558 -> 502 us,
    // 36 x 512 x 5120 x 512 878 -> 803, 64 x 256 x 2560 x 512 388 -> 356; bs=32 iteration 73.5-74.2 -> 72.0-72.1 ms.  Not for small grids: tile
    // rounds (bs=8: 21.7-22.0 -> 22.4 ms when forced) and few-column weight-gradient products (N = 256 ... 512) lose.
    static const int big = mcvc_knob("MCVC_GEMM_BIG", 1);
    if (cfg2 == 0 && big && a.N >= 1280 && (a.K % 16) == 0 && (long long)cdiv_i(a.N, 128) * (a.M / 128) * nxi >= 1280)
        return gemm2_launch<128, 128, 16, 4>(b, nxi, s);
    auto rounds_cost = [&](int bm, int bn) { return (double)cdiv_i(cdiv_i(a.N, bn) * (a.M / bm) * nxi, 256) * bm * bn; };
    static const int costsel = mcvc_knob("MCVC_GEMM_COST", 1);
    if (cfg2 == 0 && a.K >= 512 && (a.K % 32) == 0 && (a.M % 64) == 0) {
        if (!(costsel && a.ldb >= 32 && rounds_cost(128, 32) < 0.94 * rounds_cost(64, 64))) return gemm2_launch<64, 64, 32, 4>(b, nxi, s);
        return gemm2_launch<128, 32, 16, 4>(b, nxi, s);
    }
    if (narrow) {
        b.nt = cdiv_i(a.N, 32);
        static bool done = false;
        if (!done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (e != hipSuccess) return (int)e;
            done = true;
        }
        mcvc_launch(wino_gemm_kernel<32>, dim3((unsigned)(b.nt * b.mt * nxi)), dim3(256), (size_t)kGStages * (kGA + kGK * 32) * sizeof(float), s, b);
    } else {
        b.nt = cdiv_i(a.N, 64);
        mcvc_launch(wino_gemm_kernel<64>, dim3((unsigned)(b.nt * b.mt * nxi)), dim3(256), (size_t)kGStages * (kGA + kGK * 64) * sizeof(float), s, b);
    }
    return (int)hipGetLastError();
}

int mcvc_wino_input_t_launch(const WinoXformArgs& a, hipStream_t s)
{
    return xform_t_launch<0>(a, 0, 0, 4.0 * ((double)a.N * a.C * a.H * a.W + 36.0 * a.C * a.NTp), s);
}

int mcvc_wino_dy_t_launch(const WinoXformArgs& a, hipStream_t s)
{
    return xform_t_launch<1>(a, 0, 0, 4.0 * ((double)a.N * a.C * a.H * a.W + 36.0 * a.C * a.NTp), s);
}

int mcvc_wino_dw_launch(const float* du, float* dw, int Cout, int Cin, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(Cin, 256), (unsigned)Cout);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (36.0 + 50.0) * Cout * Cin);
    mcvc_launch(wino_dw_kernel, grid, dim3(256), 0, s, WinoDwKArgs{du, dw, Cout, Cin});
    return (int)hipGetLastError();
}

int mcvc_wino3_input_launch(const WinoXformArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.C);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * a.C * a.H * a.W + 16.0 * a.C * a.NT));
    mcvc_launch(wino3_input_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_wino3_output_launch(const WinoOutArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.Cout);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (16.0 * a.Cout * a.NT + 4.0 * a.Cout * a.NT));
    mcvc_launch(wino3_output_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_wino3_input_phase_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.C);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * (a.C / 4) * XH * XW + 16.0 * a.C * a.NT));
    mcvc_launch(wino3_input_phase_kernel, grid, dim3(256), 0, s, Wino3InputPhaseKArgs{a, XH, XW});
    return (int)hipGetLastError();
}

int mcvc_wino3_input_phase_t_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s)
{
    return xform_t_launch<2>(a, XH, XW, 4.0 * ((double)a.N * (a.C / 4) * XH * XW + 16.0 * a.C * a.NTp), s);
}

int mcvc_wino3_dy_t_launch(const WinoXformArgs& a, hipStream_t s)
{
    return xform_t_launch<3>(a, 0, 0, 4.0 * ((double)a.N * a.C * a.H * a.W + 16.0 * a.C * a.NTp), s);
}

int mcvc_wino3_dw_launch(const float* du, float* dw0, float* dw1, int Cout, int nbr, int Cin, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(4 * Cin, 256), (unsigned)(Cout * nbr));
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (16.0 * 4.0 + 2.0 * 25.0) * Cout * nbr * Cin);
    mcvc_launch(wino3_dw_kernel, grid, dim3(256), 0, s, Wino3DwKArgs{du, dw0, dw1, Cout, nbr, Cin});
    return (int)hipGetLastError();
}
