// Winograd F(4x4,5x5) transform kernels (wino4.h).  Every transform is two small dense products with the constant matrices; loops are fully
// unrolled, so the zero / +-1 coefficients fold away at compile time.
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include "wino4.h"

namespace {

// out = B^T in, factored through the even / odd symmetry of the points (26 operations instead of the 40 non-zero table entries: these
// kernels are instruction-bound -- 1764 instructions per (channel, tile) thread in the table-driven form, 245 us of VALU issue at 64 samples)
__device__ __forceinline__ void w4_bt(const float d[8], float o[8])
{
    const float e1 = d[2] - 4.25f * d[4] + d[6], o1 = d[1] - 4.25f * d[3] + d[5];
    const float e2 = 0.25f * d[2] - 1.25f * d[4] + d[6], o2 = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    const float e3 = 4.f * d[2] - 5.f * d[4] + d[6], o3 = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    o[0] = (d[6] - d[0]) + 5.25f * (d[2] - d[4]);
    o[1] = e1 + o1; o[2] = e1 - o1;
    o[3] = e2 + o2; o[4] = e2 - o2;
    o[5] = e3 + o3; o[6] = e3 - o3;
    o[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
}
// out = A^T m (4 from 8) and out = A d (8 from 4), the same way
__device__ __forceinline__ void w4_at(const float m[8], float o[4])
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
    o[0] = m[0] + s12 + s34 + s56;
    o[1] = d12 + 2.f * d34 + 0.5f * d56;
    o[2] = s12 + 4.f * s34 + 0.25f * s56;
    o[3] = d12 + 8.f * d34 + 0.125f * d56 + m[7];
}
__device__ __forceinline__ void w4_a(const float d[4], float o[8])
{
    const float e = d[0] + d[2], f = d[1] + d[3];
    const float e4 = d[0] + 4.f * d[2], f4 = 2.f * d[1] + 8.f * d[3];
    const float eh = d[0] + 0.25f * d[2], fh = 0.5f * d[1] + 0.125f * d[3];
    o[0] = d[0]; o[1] = e + f; o[2] = e - f; o[3] = e4 + f4; o[4] = e4 - f4; o[5] = eh + fh; o[6] = eh - fh; o[7] = d[3];
}

// V = B^T d B of the 8x8 patch whose top-left input pixel is (ih0, iw0); o[a * 8 + b]
__device__ __forceinline__ void w4_input(const float* src, int x_sh, int ih0, int iw0, int H, int W, float* o)
{
    float t[8][8];
    const float* rp = src + (long long)ih0 * x_sh + iw0;
    if (ih0 >= 0 && ih0 + 7 < H && iw0 >= 0 && iw0 + 7 < W) {       // interior window: no per-element predicates
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = rp[j];
            rp += x_sh;
            w4_bt(d, t[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float d[8];
            const int ih = ih0 + i;
            const bool rok = (ih >= 0) && (ih < H);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int iw = iw0 + j;
                d[j] = (rok && iw >= 0 && iw < W) ? rp[j] : 0.f;
            }
            rp += x_sh;
            w4_bt(d, t[i]);                      // t[i][b] = sum_j d[i][j] BT[b][j]
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float col[8], q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] = t[i][b];
        w4_bt(col, q);                           // V[a][b] = sum_i BT[a][i] t[i][b]
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) o[aa * 8 + b] = q[aa];
    }
}

// dM = A dY A^T of a 4x4 output-gradient tile (A = (A^T)^T, 8x4); o[a * 8 + b]
__device__ __forceinline__ void w4_dy(const float* src, int x_sh, int oh0, int ow0, int H, int W, float* o)
{
    float t[8][4];                               // t[a][j] = sum_i A[a][i] dy[i][j] = sum_i AT[i][a] dy[i][j]
    float dy[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dy[i][j] = (oh0 + i < H && ow0 + j < W) ? src[(long long)(oh0 + i) * x_sh + ow0 + j] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float col[4] = {dy[0][j], dy[1][j], dy[2][j], dy[3][j]};
        float q[8];
        w4_a(col, q);
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) t[aa][j] = q[aa];
    }
#pragma unroll
    for (int aa = 0; aa < 8; ++aa) w4_a(t[aa], o + aa * 8);          // o[aa][b] = sum_j t[aa][j] AT[j][b]
}

// one thread = one (channel, tile): 64 loads, 64 coalesced stores
__global__ void __launch_bounds__(256) wino4_input_kernel(const Twin<WinoXformArgs> tw)
{
    const WinoXformArgs a = tw.v[blockIdx.z];
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (tile >= a.NT) return;
    const int per = a.TH * a.TW;
    const int n = tile / per, r = tile - n * per;
    const int ty = r / a.TW, tx = r - ty * a.TW;
    float o[64];
    w4_input(a.x + (long long)n * a.x_sb + (long long)c * a.x_sc, a.x_sh, 4 * ty - a.pad, 4 * tx - a.pad, a.H, a.W, o);
    float* dst = a.v + (long long)c * a.NTp + tile;
    const long long xs = (long long)a.C * a.NTp;
#pragma unroll
    for (int q = 0; q < 64; ++q) dst[(long long)q * xs] = o[q];
}

__global__ void __launch_bounds__(256) wino4_output_kernel(const Twin<WinoOutArgs> tw)
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
    float u[4][8];                               // u[i][b] = sum_a AT[i][a] M[a][b]
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float col[8];
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) col[aa] = src[(long long)(aa * 8 + b) * xs];
        float q[4];
        w4_at(col, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i][b] = q[i];
    }
    const float bias = a.bias ? a.bias[co] : 0.f;
    float* yn = a.y + (long long)n * a.y_sb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float row[4];
        w4_at(u[i], row);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float acc = row[j] + bias;
            const int oh = 4 * ty + i, ow = 4 * tx + j;
            if (oh >= a.OH || ow >= a.OW) continue;
            long long off;
            if (a.shuffle) {
                const int yh = 2 * oh + ((co >> 1) & 1), yw = 2 * ow + (co & 1);
                if (yh >= a.YH || yw >= a.YW) continue;
                off = (long long)(co >> 2) * a.y_sc + (long long)yh * a.y_sh + yw;
            } else {
                off = (long long)co * a.y_sc + (long long)oh * a.y_sh + ow;
            }
            if (a.accumulate) yn[off] += acc; else yn[off] = acc;
        }
    }
}

// tile-major operands of the weight gradient through an LDS transpose: 4 tiles x 64 channels per workgroup (see xform_t_kernel in
// wino_kernels.hip); KIND 0: V^T from x, 1: dM^T from dY
constexpr int kX4T = 4, kX4C = 64, kX4Pitch = 68;
template <int KIND>
__global__ void __launch_bounds__(256) xform4_t_kernel(const Twin<WinoXformArgs> tw)
{
    const WinoXformArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float xbuf[];          // [64][kX4T][kX4Pitch]
    const int tid = threadIdx.x;
    const int tile0 = blockIdx.x * kX4T, c0 = blockIdx.y * kX4C;
    {
        const int tl = tid & 3, cl = tid >> 2;
        const int tile = tile0 + tl, c = c0 + cl;
        float o[64];
        if (tile < a.NT && c < a.C) {
            const int per = a.TH * a.TW;
            const int n = tile / per, r = tile - n * per;
            const int ty = r / a.TW, tx = r - ty * a.TW;
            const float* src = a.x + (long long)n * a.x_sb + (long long)c * a.x_sc;
            if constexpr (KIND == 0) w4_input(src, a.x_sh, 4 * ty - a.pad, 4 * tx - a.pad, a.H, a.W, o);
            else w4_dy(src, a.x_sh, 4 * ty, 4 * tx, a.H, a.W, o);
        } else {
#pragma unroll
            for (int q = 0; q < 64; ++q) o[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 64; ++q) xbuf[(q * kX4T + tl) * kX4Pitch + cl] = o[q];
    }
    __syncthreads();
    const int cl = tid & 63, tl = tid >> 6, c = c0 + cl, tile = tile0 + tl;
    if (c < a.C && tile < a.NTp) {
#pragma unroll 8
        for (int q = 0; q < 64; ++q) a.v[((long long)q * a.NTp + tile) * a.C + c] = xbuf[(q * kX4T + tl) * kX4Pitch + cl];
    }
}

// dg = G^T dU G, accumulated into the OIHW gradient.  One thread per (co, ci), ci fastest (coalesced reads of dU).
struct Wino4DwKArgs { const float* du; float* dw; int Cout; int Cin; };
__global__ void __launch_bounds__(256) wino4_dw_kernel(const Twin<Wino4DwKArgs> tw)
{
    const Wino4DwKArgs a = tw.v[blockIdx.z];
    __shared__ float wt[256 * 25];               // (coalesced read-modify-write of dW through LDS: see wino_dw_kernel)
    const int ci = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    const bool live = ci < a.Cin;
    const long long xs = (long long)a.Cout * a.Cin;
    const float* src = a.du + (long long)co * a.Cin + (live ? ci : 0);
    float t[5][8];                               // t[k][b] = sum_a G[a][k] dU[a][b]
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float col[8];
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) col[aa] = src[(long long)(aa * 8 + b) * xs];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int aa = 0; aa < 8; ++aa) acc += kW4G[aa][k] * col[aa];
            t[k][b] = acc;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            float acc = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b) acc += t[k][b] * kW4G[b][l];
            wt[threadIdx.x * 25 + k * 5 + l] = acc;
        }
    __syncthreads();
    int nci = a.Cin - (int)blockIdx.x * 256; if (nci > 256) nci = 256;
    float* dst = a.dw + ((long long)co * a.Cin + (long long)blockIdx.x * 256) * 25;
    float old[25];                               // (all loads of the read-modify-write before the first store)
#pragma unroll
    for (int u = 0; u < 25; ++u) { const int i = threadIdx.x + u * 256; old[u] = (i < nci * 25) ? dst[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 25; ++u) { const int i = threadIdx.x + u * 256; if (i < nci * 25) dst[i] = old[u] + wt[i]; }
}

}  // namespace

int mcvc_wino4_input_launch(const WinoXformArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.C);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * a.C * a.H * a.W + 64.0 * a.C * a.NT));
    mcvc_launch(wino4_input_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_wino4_output_launch(const WinoOutArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.NT, 256), (unsigned)a.Cout);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (64.0 * a.Cout * a.NT + 16.0 * a.Cout * a.NT));
    mcvc_launch(wino4_output_kernel, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

template <int KIND>
static int xform4_t_launch(const WinoXformArgs& a, hipStream_t s)
{
    constexpr size_t lds = (size_t)64 * kX4T * kX4Pitch * sizeof(float);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xform4_t_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    dim3 grid((unsigned)cdiv_i(a.NTp, kX4T), (unsigned)cdiv_i(a.C, kX4C));
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * ((double)a.N * a.C * a.H * a.W + 64.0 * a.C * a.NTp));
    mcvc_launch(xform4_t_kernel<KIND>, grid, dim3(256), lds, s, a);
    return (int)hipGetLastError();
}
int mcvc_wino4_input_t_launch(const WinoXformArgs& a, hipStream_t s) { return xform4_t_launch<0>(a, s); }
int mcvc_wino4_dy_t_launch(const WinoXformArgs& a, hipStream_t s) { return xform4_t_launch<1>(a, s); }

int mcvc_wino4_dw_launch(const float* du, float* dw, int Cout, int Cin, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(Cin, 256), (unsigned)Cout);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (64.0 + 50.0) * Cout * Cin);
    mcvc_launch(wino4_dw_kernel, grid, dim3(256), 0, s, Wino4DwKArgs{du, dw, Cout, Cin});
    return (int)hipGetLastError();
}
