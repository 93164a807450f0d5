// Winograd F(4x4, 5x5) for the stride-1 5x5 convolutions (upSample1 / upSample2) at LARGER batches: 8x8 input tiles, 64 transform points
// per 16 outputs = 4 multiplies per output against 9 for F(2x2,5x5) and 25 for the direct form; V and M are 4x the activation instead of
// 9x.  Interpolation points {0, +-1, +-2, +-1/2, inf} (Cook-Toom, un-normalised): fp32 error of a 256-channel product 4.7e-6 relative,
// the same as the F(2x2,5x5) scheme in use (4.4e-6; measured on the CPU with this construction), three orders inside the 1e-3 bar.
// At one or two samples per pass the 36-point form stays: its products are weight-streaming bound and the 64-point weight sets are
// 1.78x larger.  Shares WinoXformArgs / WinoOutArgs / the batched GEMM with wino.h (TH, TW count 4x4 tiles here).
#pragma once
#include <hip/hip_runtime.h>
#include "wino.h"

// A^T (4x8), G (8x5), B^T (8x8)
__device__ constexpr float kW4AT[4][8] = {
    {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
    {0.f, 1.f, -1.f, 2.f, -2.f, 0.5f, -0.5f, 0.f},
    {0.f, 1.f, 1.f, 4.f, 4.f, 0.25f, 0.25f, 0.f},
    {0.f, 1.f, -1.f, 8.f, -8.f, 0.125f, -0.125f, 1.f}};
__device__ constexpr float kW4G[8][5] = {
    {-1.f, 0.f, 0.f, 0.f, 0.f},
    {-2.f / 9.f, -2.f / 9.f, -2.f / 9.f, -2.f / 9.f, -2.f / 9.f},
    {-2.f / 9.f, 2.f / 9.f, -2.f / 9.f, 2.f / 9.f, -2.f / 9.f},
    {1.f / 90.f, 1.f / 45.f, 2.f / 45.f, 4.f / 45.f, 8.f / 45.f},
    {1.f / 90.f, -1.f / 45.f, 2.f / 45.f, -4.f / 45.f, 8.f / 45.f},
    {32.f / 45.f, 16.f / 45.f, 8.f / 45.f, 4.f / 45.f, 2.f / 45.f},
    {32.f / 45.f, -16.f / 45.f, 8.f / 45.f, -4.f / 45.f, 2.f / 45.f},
    {0.f, 0.f, 0.f, 0.f, 1.f}};
__device__ constexpr float kW4BT[8][8] = {
    {-1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f, 1.f, 0.f},
    {0.f, 1.f, 1.f, -4.25f, -4.25f, 1.f, 1.f, 0.f},
    {0.f, -1.f, 1.f, 4.25f, -4.25f, -1.f, 1.f, 0.f},
    {0.f, 0.5f, 0.25f, -2.5f, -1.25f, 2.f, 1.f, 0.f},
    {0.f, -0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f, 0.f},
    {0.f, 2.f, 4.f, -2.5f, -5.f, 0.5f, 1.f, 0.f},
    {0.f, -2.f, 4.f, 2.5f, -5.f, -0.5f, 1.f, 0.f},
    {0.f, -1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f, 1.f}};

int mcvc_wino4_input_launch(const WinoXformArgs& a, hipStream_t s);      // x -> V[64][C][NTp]
int mcvc_wino4_output_launch(const WinoOutArgs& a, hipStream_t s);       // M[64][Cout][NTp] -> y (+bias, PixelShuffle store, accumulate)
int mcvc_wino4_input_t_launch(const WinoXformArgs& a, hipStream_t s);    // x -> Vt[64][NTp][C]
int mcvc_wino4_dy_t_launch(const WinoXformArgs& a, hipStream_t s);       // dY -> dMt[64][NTp][C] = A dY A^T
int mcvc_wino4_dw_launch(const float* du, float* dw, int Cout, int Cin, hipStream_t s);   // dw[co][ci][5][5] += G^T dU G, dU[64][Cout][Cin]

// Weight transform U = G g G^T (8x8 from 5x5), one thread per (co, ci), called from the whole-network re-pack kernel; same conventions as
// wino_weight_tile (dgrad: taps flipped, rows = output channels).
static __device__ __forceinline__ void wino4_weight_core(const float* g, float* dst, int co, int ci, int ld, long long xi_stride, int co_off, int dgrad)
{
    float t[5][8];                                  // t[k][b] = sum_l g[k][l] G[b][l]
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float gr[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) gr[l] = dgrad ? g[(4 - k) * 5 + (4 - l)] : g[k * 5 + l];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float acc = 0.f;
#pragma unroll
            for (int l = 0; l < 5; ++l) acc += gr[l] * kW4G[b][l];
            t[k][b] = acc;
        }
    }
    const long long row = dgrad ? (co_off + co) : ci;
    const long long col = dgrad ? ci : (co_off + co);
    float* d0 = dst + row * ld + col;
#pragma unroll
    for (int aa = 0; aa < 8; ++aa)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) acc += kW4G[aa][k] * t[k][b];
            d0[(long long)(aa * 8 + b) * xi_stride] = acc;
        }
}
static __device__ __forceinline__ void wino4_weight_tile(const float* w, float* dst, int Cout, int Cin, int ld, long long xi_stride, int co_off, int dgrad, int bx, int by)
{
    const int a_idx = bx * 256 + threadIdx.x, b_idx = by;
    const int co = dgrad ? b_idx : a_idx, ci = dgrad ? a_idx : b_idx;
    if (co >= Cout || ci >= Cin) return;
    wino4_weight_core(w + ((long long)co * Cin + ci) * 25, dst, co, ci, ld, xi_stride, co_off, dgrad);
}
