// GEMM forms of the convolutions that are plain matrix products, on the 64x64 / k32 LDS-DMA GEMM pipeline (wino_kernels.hip) -- all of them
// IMPLICIT since r5: every operand is read where its producer left it, no tap planes, no transposed copies, no gather kernels (the staged
// forms of r2-r4 -- im2col_s2 / im2col_s2_t / col2im_s2 / im2col_1d / im2col_1d_t / col2im_1d / planes_t -- are gone):
//
//   * the discriminators' 3 x 3 stride-2 layers (downSample1-3, model.py:298-314): forward and data gradient gather their B operand from
//     the phase-split padded input / the padded dY (igemm_kernel), the data gradient reads the forward weight copy row-major (arow), the
//     weight gradient reads both operands in place (wgemm_kernels.hip);
//   * the 1 x KW convolutions of the 1-D trunk beyond the fused small-batch kernels (more than 64 columns, model.py:47-76, 142-189): the
//     same two kernels over DENSE rows -- a tap's window is the row shifted by -1 / 0 / +1 column, and the value a shifted window picks up
//     from the neighbouring row at a row's end is dropped at the operand read (zw / zs below); a 1 x 1 convolution multiplies the activation
//     itself (sgemm_kernel).
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

// ---- implicit GEMM (r4): the B operand is gathered from the activation itself, no tap planes in HBM --------------------------------------
// One launch = up to 4 products ("classes": blockIdx.y) that share B, C and the geometry.  A class is  C_c[m][n] = sum_{t, c} A_c[t*Cb + c][m] *
// B(t, c, n)  with a per-tap shift of the gathered window:
//     B(t, c, n = (bb, i, j)) = b[boff_c[t] + c*b_cs + bb*b_sn + i*b_pitch + j]        (i, j = row / column of n inside its image: n % P = i*OW + j)
//     C_c(m, n)               = c[coff_c + m*ldc + bb*c_sn + i*c_sh + j*c_sw]
// A stage of 32 k lies inside one tap (Cb % 32 == 0), so its B rows are 32 consecutive channels at one uniform offset: the same 16-byte
// LDS-DMA pieces as the plain GEMM, at addresses that are only 4-byte aligned where a tap shifts the window by one column (measured: fine).
//   forward of a 3x3 stride-2 convolution: one class, 9 taps over the PHASE-SPLIT padded input (xs layout below), C = y in place;
//   data gradient: the four output-parity classes (1 + 2 + 2 + 4 taps) over dY with one zero column / row of padding, C scattered to
//   dx[2a + qh][2b + qw] -- every input pixel is written by exactly one class: no tap planes, no gather kernel, no atomics.
// A of tap t: element (c, m) at a[aoff[t] + c*a_ks + m]  (K-major rows, m contiguous; a_ks = floats between consecutive k rows), or with
// arow = 1 at a[aoff[t] + m*a_ks + c] -- a ROW-major source, the 32 k of a stage contiguous: the forward weight copy of a convolution serving
// its data gradient (m = input channel, c = output channel).  zs[t] (with IGemmArgs.zw): 1 / 2 = tap t's window is shifted by -1 / +1 column
// over dense rows of zw columns -- B values at a row's first / last column are taken as the zero the padding holds.
struct IGemmClass { const float* a; int ntaps; long long coff; long long boff[9]; long long aoff[9]; int zs[9]; };
struct IGemmArgs {
    IGemmClass cls[4]; int ncls;
    long long a_ks; int zw;
    const float* b; long long b_cs, b_sn; int b_pitch;
    int Cb;                             // channels per tap; K of class c = cls[c].ntaps * Cb
    int OW, P;                          // n -> (bb = n / P, i = (n % P) / OW, j = n % OW);  OW % 4 == 0
    float* c; long long ldc, c_sn; int c_sh, c_sw;
    const float* bias; int accumulate; int arow;
    int M, N;                           // M % 64 == 0, N % 4 == 0
    int nsplit; float* c_slab; long long c_split;      // K split of every class: split s > 0 writes the C layout at c_slab + (s - 1) * c_split
    int nt, mt, mgroup;                 // (filled by the launcher; mgroup: row tiles per L2-sized group, see igemm_kernel)
};
int mcvc_igemm_launch(const IGemmArgs& a, hipStream_t s);

// ---- implicit weight gradient (r5, wgemm_kernels.hip): both operands read where they lie, pixel-contiguous rows, no transposed copies ----------
//     dW[co][ci][t] (+)= sum_{n = (bb, i, j)} dY(co, n) * X_t(ci, n)
//     dY(co, n)  = a[co*a_cs + bb*a_sn + i*a_pitch + j]                   (dense planes or the padded dY layout)
//     X_t(ci, n) = b[boff[t] + ci*b_cs + bb*b_sn + i*b_pitch + j]         (tap t's window: of the phase-split padded input for the 3 x 3 stride-2 layers)
// dW = the OIHW gradient tensor(s): rows [0, m_split) -> c, the rest -> c2 (value | gate), row pitch Cin * ntaps.  K split over the pixels:
// split s > 0 writes the same layout at c_slab + (s - 1) * c_split (split 0 then writes c plain; mcvc_dw_accum_launch sums), nsplit == 1 adds
// straight into the gradient when `accumulate`.
struct WGemmArgs {
    const float* a; long long a_cs, a_sn; int a_pitch;
    const float* b; long long b_cs, b_sn; int b_pitch; long long boff[9]; int ntaps;      // ntaps = 9 (3 x 3), 3 or 1 (1-D)
    int zw;                              // ntaps = 3 over DENSE rows of zw columns (the 1-D trunk): taps 0 / 2 are shifted by -1 / +1 column
                                         // (boff = -1 / +1) and their x values at a row's first / last column count as the padding's zero
    int OW, P, NPIX;                     // n -> (bb = n / P, i = (n % P) / OW, j = n % OW); OW % 4 == 0; NPIX = samples * P
    int M, Cin;                          // M % 128 == 0; Cin % 32 == 0 (ntaps = 9) / % 64 == 0
    float* c; float* c2; int m_split; int accumulate;
    int nsplit; float* c_slab; long long c_split;
    long long ldc; int nstages, nt, mt;  // (filled by the launcher)
    int xcd_order;                       // (filled by the launcher) 1: every XCD gets one contiguous range of tiles (see wgemm_kernel)
};
int mcvc_wgemm_launch(const WGemmArgs& a, hipStream_t s);
int mcvc_wgemm_cib(int taps);            // input channels per workgroup tile

// Phase-split padded activation layout ("xs") read by the implicit forward GEMM of a 3x3 stride-2 padding-1 convolution over an H x W image
// (H, W even): per (sample, channel) four planes pq = 2*(h & 1) + (w & 1), each (H/2 + 1) rows of PW = W/2 + 4 floats; element (h, w) sits at
// row (h >> 1) + 1, column (w >> 1) + 4; row 0 and columns 0..3 are zero (the taps kh = 0 / kw = 0 of the first output row / column read them).
static inline int mcvc_xs_pw(int W) { return W / 2 + 4; }
static inline long long mcvc_xs_plane(int H, int W) { return (long long)(H / 2 + 1) * (W / 2 + 4); }
static inline long long mcvc_xs_floats(int C, int H, int W) { return 4LL * C * mcvc_xs_plane(H, W); }       // per sample
// dY layout read by the implicit data gradient: planes of (OH + 1) rows x (OW + 4) floats, zero beyond row OH - 1 / column OW - 1
static inline int mcvc_dyp_pitch(int OW) { return OW + 4; }
static inline long long mcvc_dyp_plane(int OH, int OW) { return (long long)(OH + 1) * (OW + 4); }

// dense [NB][C][H][W] -> the phase-split padded layout (zero borders included) / dense dY [NB][C][OH][OW] -> the padded dY layout: the
// networks' producers write these layouts themselves; these two kernels serve the op-level entries (mcvc_layer_*, scheme 5), whose
// operands arrive dense
int mcvc_xs_from_dense_launch(const float* x, float* xs, int NB, int C, int H, int W, hipStream_t s);
int mcvc_dyp_from_dense_launch(const float* dy, float* dyp, int NB, int C, int OH, int OW, hipStream_t s);

// g0[co][k] += sum_s slabs[s][co][k] (co < Cout), g1[co - Cout][k] += ... (co >= Cout; g1 may be null when rows == Cout)
int mcvc_dw_accum_launch(const float* slabs, int nslab, long long slab_stride, float* g0, float* g1, int Cout, int rows, int K9, hipStream_t s);
