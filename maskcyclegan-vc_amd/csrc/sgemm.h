// Strided 3x3 convolutions of the discriminators (downSample1-3, model.py:298-314) as staged GEMMs.
//
// The three stride-2 layers are plain matrix products, and the 64x64 / k32 LDS-DMA GEMM pipeline (wino_kernels.hip) runs them at
// 105-120 TF/s at 32+ samples per pass where the direct stride-2 kernels reach 40-60 (forward), 40 (data gradient) and 35-39 TF/s
// (weight gradient): profiles/r02_gemm_probe.log; with a K-split it also wins at one sample per pass (13 % of the bs=1 iteration).
// The operands that are not already matrices are staged once per pass (tap planes of x, 2.25x its size; all of it stays in HBM / MALL):
//
//   forward        Y[co][n]     = sum_k  Wt[k][co]  * Xcol[k][n]        k = 9*ci + 3*kh + kw, n = (b, oh, ow); + bias, stored as y[b][co][p]
//   data gradient  dXcol[k][n]  = sum_co W[co][k]   * dY[co][n]         W = the OIHW parameter tensors themselves (value | gate rows),
//                                                                      dY read in place ([b][co][p]); then dx[b][ci][ih][iw] gathers its <= 4 taps
//   weight grad.   dW[co][k]    = sum_n  dYt[n][co] * XcolT[n][k]       both operands pixel-major; K-split slabs, then dw += sum of slabs
//
#pragma once
#include <hip/hip_runtime.h>

struct SGemmArgs {
    // A: [K][M] (M contiguous).  Rows [0, k_split) come from a, rows [k_split, K) from a2 (its row 0 = row k_split); k_split = K: a only
    const float* a; const float* a2; int k_split; long long lda;
    // B: [K][N]:  B(k, n) = b[k*ldb + (n / bseg) * b_sn + n % bseg]      (bseg = N, b_sn = 0: a plain matrix)
    const float* b; long long ldb; int bseg; long long b_sn;
    // C: [M][N]:  C(m, n) = (m < m_split ? c : c2)[m' * ldc + (n / cseg) * c_sn + n % cseg], m' = m (c) or m - m_split (c2)
    float* c; float* c2; int m_split; long long ldc; int cseg; long long c_sn;
    const float* bias;        // [M] added to every column, or nullptr
    int accumulate;           // C += (split 0; no atomics: one workgroup owns a tile)
    int M, N, K;              // M % 64 == 0, K % (32 * nsplit) == 0, N % 4 == 0
    // K-split: split s handles rows [s*K/nsplit, (s+1)*K/nsplit); split 0 writes C (+ bias), split s > 0 writes the same layout at
    // c_slab + (s - 1) * c_split (c2 / m_split apply to split 0 only; use them with nsplit = 1)
    int nsplit; float* c_slab; long long c_split;
    int nt, mt;               // (filled by the launcher)
};
int mcvc_sgemm_launch(const SGemmArgs& a, hipStream_t s);

struct StageArgs {
    const float* x; long long x_sb, x_sc; int x_sh;      // image view [NB][C][H][W] (W contiguous)
    int NB, C, H, W, OH, OW;                               // 3x3, stride 2, padding 1: OH = (H + 1) / 2 ...
    float* out; long long ld;                              // see the launchers
    int rows_pad;                                          // transposed forms: rows [NB*OH*OW, rows_pad) are written as zeros
};
// Xcol[k = 9*ci + tap][n], ld >= NB*OH*OW
int mcvc_im2col_s2_launch(const StageArgs& a, hipStream_t s);
// XcolT[n][k], ld = 9*C
int mcvc_im2col_s2_t_launch(const StageArgs& a, hipStream_t s);
// Yt[n][c] from y[b][c][p] (H x W = the plane of y; OH/OW unused), ld = C
int mcvc_planes_t_launch(const StageArgs& a, hipStream_t s);
// dx[b][ci][ih][iw] (=|+=) sum over the taps that reach it of dXcol[9*ci + tap][n], summed over nslab K-split slabs of dXcol
// (x / x_* = the dx view; out = dXcol, read)
int mcvc_col2im_s2_launch(const StageArgs& a, int nslab, long long slab_stride, int accumulate, hipStream_t s);
// The same for 1 x KW convolutions along w (KW = 1, 3; stride 1, padding (KW-1)/2) over an image of NB*H rows -- the 1-D trunk at more
// than 32 columns (model.py:47-76, 142-189): k = KW*ci + tap, n = (b*H + h)*W + w
int mcvc_im2col_1d_launch(const StageArgs& a, int KW, hipStream_t s);
int mcvc_im2col_1d_t_launch(const StageArgs& a, int KW, hipStream_t s);
int mcvc_col2im_1d_launch(const StageArgs& a, int KW, int nslab, long long slab_stride, int accumulate, hipStream_t s);
// g0[co][k] += sum_s slabs[s][co][k] (co < Cout), g1[co - Cout][k] += ... (co >= Cout; g1 may be null when rows == Cout)
int mcvc_dw_accum_launch(const float* slabs, int nslab, long long slab_stride, float* g0, float* g1, int Cout, int rows, int K9, hipStream_t s);
