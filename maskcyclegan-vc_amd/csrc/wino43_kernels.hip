// Winograd F(4x4,3x3) for the stride-2 5x5 convolutions in phase form (downSample1 / downSample2 of the generators, model.py:180-190) at
// larger batch.  The phase formulation (wino.h: wino3_*) turns the stride-2 5x5 conv into a 3x3 stride-1 conv over 4*Cin phase planes (and
// its data gradient into a 3x3 conv over dY with 4*Cin parity-class outputs); F(2x2,3x3) spends 16 multiplies per 4 outputs there, this
// variant 36 per 16 (1.78x fewer) on 6x6 windows at stride 4 with the points {0, +-1, +-2, inf} -- the same B^T as F(2x2,5x5), the first
// three columns of its G, and two more rows of its A^T.  Operand layouts are those of the 16-point kernels with 36 matrices, so the batched
// GEMM (wino_gemm / gemm2, nxi = 36) is shared.  Needs output sizes that are multiples of 4 (frames % 16 == 0).
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include "wino.h"

namespace {

__device__ __forceinline__ void bt6(const float d[6], float o[6])
{
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A^T (4x6) = rows point^i:  y = A^T m
__device__ __forceinline__ void at46(const float m[6], float o[4])
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}
// A (6x4):  dM = A dy A^T
__device__ __forceinline__ void a64(const float d[4], float o[6])
{
    const float e = d[0] + d[2], f = d[1] + d[3], e4 = d[0] + 4.f * d[2], f4 = 2.f * d[1] + 8.f * d[3];
    o[0] = d[0]; o[1] = e + f; o[2] = e - f; o[3] = e4 + f4; o[4] = e4 - f4; o[5] = d[3];
}

// 6x6 window of (channel c, tile) -> t = B^T d B in o[aa * 6 + b].  PHASE: channel k = 4ci + 2p + q is the plane x[ci][2i+p][2j+q]
template <bool PHASE>
__device__ __forceinline__ void window36(const WinoXformArgs& a, int XH, int XW, int tile, int c, float* o)
{
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    const int i0 = 4 * ty - a.pad, j0 = 4 * tx - a.pad;
    const int ci = PHASE ? (c >> 2) : c, p = (c >> 1) & 1, q = c & 1;
    const float* src = a.x + (long long)n * a.x_sb + (long long)ci * a.x_sc;
    float t[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float d[6];
        const int ih = PHASE ? 2 * (i0 + i) + p : i0 + i;
        const bool rok = (i0 + i >= 0) && (ih < (PHASE ? XH : a.H));
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int iw = PHASE ? 2 * (j0 + j) + q : j0 + j;
            d[j] = (rok && j0 + j >= 0 && iw < (PHASE ? XW : a.W)) ? src[(long long)ih * a.x_sh + iw] : 0.f;
        }
        bt6(d, t[i]);
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float col[6], w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = t[i][b];
        bt6(col, w);
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) o[aa * 6 + b] = w[aa];
    }
}

struct W43InKArgs { WinoXformArgs a; int XH; int XW; };
template <bool PHASE>
__global__ void __launch_bounds__(256) wino43_input_kernel(const Twin<W43InKArgs> tw)
{
    const W43InKArgs ka_ = tw.v[blockIdx.z];
    const WinoXformArgs& a = ka_.a;
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (tile >= a.NT) return;
    float o[36];
    window36<PHASE>(a, ka_.XH, ka_.XW, tile, c, o);
    float* dst = a.v + (long long)c * a.NTp + tile;
    const long long xs = (long long)a.C * a.NTp;
#pragma unroll
    for (int q = 0; q < 36; ++q) dst[(long long)q * xs] = o[q];
}

// M[36][Cout][tile] -> 4x4 outputs per tile (+ bias); `shuffle`: channel co = 4c + 2qh + qw is the parity class (qh, qw) of plane c
__global__ void __launch_bounds__(256) wino43_output_kernel(const Twin<WinoOutArgs> tw)
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
    float u[4][6];                               // u[i][b] = sum_a A^T[i][a] m[a][b]
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float m[6], w[4];
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) m[aa] = src[(long long)(aa * 6 + b) * xs];
        at46(m, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i][b] = w[i];
    }
    const float bias = a.bias ? a.bias[co] : 0.f;
    float* yn = a.y + (long long)n * a.y_sb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ov[4];
        at46(u[i], ov);
        const int oh = 4 * ty + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ow = 4 * tx + j;
            if (oh >= a.OH || ow >= a.OW) continue;
            long long off;
            if (a.shuffle) {
                const int yh = 2 * oh + ((co >> 1) & 1), yw = 2 * ow + (co & 1);
                if (yh >= a.YH || yw >= a.YW) continue;
                off = (long long)(co >> 2) * a.y_sc + (long long)yh * a.y_sh + yw;
            } else {
                off = (long long)co * a.y_sc + (long long)oh * a.y_sh + ow;
            }
            const float v = ov[j] + bias;
            if (a.accumulate) yn[off] += v; else yn[off] = v;
        }
    }
}

// ---- weight-gradient operands, tile-major: [36][NTp][C], rows of tiles >= NT zero.  8 tiles x 64 channels per workgroup through LDS so that
// reads walk the tiles of a row and stores are 256-byte runs of channels (the layout of xform_t_kernel, wino_kernels.hip).
//   KIND 0: V^T of the phase-plane input      KIND 1: dM^T = A dY A^T from the 4x4 output-gradient tiles
constexpr int kXT = 8, kXC = 64, kXPitch = 68;
template <int KIND>
__global__ void __launch_bounds__(256) xform43_t_kernel(const Twin<W43InKArgs> tw)
{
    const W43InKArgs ka_ = tw.v[blockIdx.z];
    const WinoXformArgs& a = ka_.a;
    extern __shared__ __attribute__((aligned(16))) float xbuf[];          // [36][kXT][kXPitch]
    const int tid = threadIdx.x;
    const int tile0 = blockIdx.x * kXT, c0 = blockIdx.y * kXC;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int tl = tid & 7, cl = (tid >> 3) + 32 * it;
        const int tile = tile0 + tl, c = c0 + cl;
        float o[36];
        if (tile < a.NT && c < a.C) {
            if constexpr (KIND == 0) window36<true>(a, ka_.XH, ka_.XW, tile, c, o);
            else {
                const int per = a.TH * a.TW;
                const int n = tile / per, r = tile - n * per;
                const int ty = r / a.TW, tx = r - ty * a.TW;
                const float* src = a.x + (long long)n * a.x_sb + (long long)c * a.x_sc;
                float t[4][6];                   // t[j][aa] = sum_i A[aa][i] dy[i][j]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int oh = 4 * ty + i, ow = 4 * tx + j;
                        d[i] = (oh < a.H && ow < a.W) ? src[(long long)oh * a.x_sh + ow] : 0.f;
                    }
                    a64(d, t[j]);
                }
#pragma unroll
                for (int aa = 0; aa < 6; ++aa) {
                    const float d[4] = {t[0][aa], t[1][aa], t[2][aa], t[3][aa]};
                    a64(d, o + aa * 6);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 36; ++q) o[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 36; ++q) xbuf[(q * kXT + tl) * kXPitch + cl] = o[q];
    }
    __syncthreads();
    const int cl = tid & 63, c = c0 + cl;
    if (c < a.C) {
#pragma unroll 4
        for (int q = 0; q < 36; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tl = (tid >> 6) + 4 * h, tile = tile0 + tl;
                if (tile < a.NTp) a.v[((long long)q * a.NTp + tile) * a.C + c] = xbuf[(q * kXT + tl) * kXPitch + cl];
            }
    }
}

template <int KIND>
int xform43_t_launch(const WinoXformArgs& a, int XH, int XW, double bytes, hipStream_t s)
{
    constexpr size_t lds = (size_t)36 * kXT * kXPitch * sizeof(float);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xform43_t_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    dim3 grid((unsigned)cdiv_i(a.NTp, kXT), (unsigned)cdiv_i(a.C, kXC));
    TraceScope ts(K_ELEMENTWISE, s, 0.0, bytes);
    mcvc_launch(xform43_t_kernel<KIND>, grid, dim3(256), lds, s, W43InKArgs{a, XH, XW});
    return (int)hipGetLastError();
}

// dg' = G^T dU G (3x3 per (co, k = 4ci+2p+q), G = 6x3), scattered into the OIHW gradients: dw[co][ci][2u'+p][2v'+q] += dg'[u'][v']
struct W43DwKArgs { const float* du; float* dw0; float* dw1; int Cout; int nbr; int Cin; };
__global__ void __launch_bounds__(256) wino43_dw_kernel(const Twin<W43DwKArgs> tw)
{
    const W43DwKArgs ka_ = tw.v[blockIdx.z];
    const int Cout = ka_.Cout, nbr = ka_.nbr, Cin = ka_.Cin;
    const int k = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    const int K = 4 * Cin;
    if (k >= K) return;
    const long long xs = (long long)Cout * nbr * K;
    const float* src = ka_.du + (long long)co * K + k;
    auto gt = [](const float v[6], float o[3]) {
        const float s12 = v[1] + v[2], d12 = v[1] - v[2], s34 = v[3] + v[4], d34 = v[3] - v[4];
        o[0] = 0.25f * v[0] - s12 * (1.0f / 6.0f) + s34 * (1.0f / 24.0f);
        o[1] = -d12 * (1.0f / 6.0f) + d34 * (1.0f / 12.0f);
        o[2] = -s12 * (1.0f / 6.0f) + s34 * (1.0f / 6.0f) + v[5];
    };
    float t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float col[6], o[3];
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) col[aa] = src[(long long)(aa * 6 + b) * xs];
        gt(col, o);
#pragma unroll
        for (int u = 0; u < 3; ++u) t[u][b] = o[u];
    }
    const int ci = k >> 2, p = (k >> 1) & 1, q = k & 1;
    float* dw = (co < Cout) ? ka_.dw0 : ka_.dw1;
    const int col = (co < Cout) ? co : co - Cout;
    if (!dw) return;
    float* dst = dw + ((long long)col * Cin + ci) * 25;
    // (the nine read-modify-writes: all loads before the first store -- written as `dst[..] += o` they run as nine sequential round trips)
    float r[3][3], old[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u) gt(t[u], r[u]);
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

}  // namespace

int mcvc_wino43_input_launch(const WinoXformArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.C);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * a.C * a.H * a.W + 36.0 * a.C * a.NT));
    mcvc_launch(wino43_input_kernel<false>, grid, dim3(256), 0, s, W43InKArgs{a, 0, 0});
    return (int)hipGetLastError();
}

int mcvc_wino43_input_phase_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.C);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * (a.C / 4) * XH * XW + 36.0 * a.C * a.NT));
    mcvc_launch(wino43_input_kernel<true>, grid, dim3(256), 0, s, W43InKArgs{a, XH, XW});
    return (int)hipGetLastError();
}

int mcvc_wino43_output_launch(const WinoOutArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.Cout);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (36.0 * a.Cout * a.NT + 16.0 * a.Cout * a.NT));
    mcvc_launch(wino43_output_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_wino43_input_phase_t_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s)
{
    return xform43_t_launch<0>(a, XH, XW, 4.0 * ((double)a.N * (a.C / 4) * XH * XW + 36.0 * a.C * a.NTp), s);
}

int mcvc_wino43_dy_t_launch(const WinoXformArgs& a, hipStream_t s)
{
    return xform43_t_launch<1>(a, 0, 0, 4.0 * ((double)a.N * a.C * a.H * a.W + 36.0 * a.C * a.NTp), s);
}

int mcvc_wino43_dw_launch(const float* du, float* dw0, float* dw1, int Cout, int nbr, int Cin, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(4 * Cin, 256), (unsigned)(Cout * nbr));
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (36.0 * 4.0 + 2.0 * 25.0) * Cout * nbr * Cin);
    mcvc_launch(wino43_dw_kernel, grid, dim3(256), 0, s, W43DwKArgs{du, dw0, dw1, Cout, nbr, Cin});
    return (int)hipGetLastError();
}
