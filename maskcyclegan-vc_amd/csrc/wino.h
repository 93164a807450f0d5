// Winograd F(2x2, 5x5) pieces for the stride-1 5x5 convolutions (upSample1 / upSample2 forward and data-gradient).
#pragma once
#include <hip/hip_runtime.h>

struct WinoXformArgs {
    // input transform: x[n][c][h][w] -> V[xi][c][tile], xi = 0..35, tile = (n*TH + ty)*TW + tx
    const float* x; long long x_sb, x_sc; int x_sh;
    float* v;                 // [36][C][NT]
    int N, C, H, W;           // input image
    int TH, TW, NT;           // tiles per image (rows, cols) and in total
    int NTp;                  // tile pitch of V (NT rounded up to 32)
    int pad;                  // conv padding (2)
};

struct WinoOutArgs {
    // output transform: M[xi][co][tile] -> y (+bias), 2x2 outputs per tile; optional PixelShuffle(2) store
    const float* m;           // [36][Cout][NT]
    const float* bias;        // [Cout] or nullptr
    float* y; long long y_sb, y_sc; int y_sh;
    int N, Cout, OH, OW, TH, TW, NT, NTp;
    int shuffle, YH, YW;
    int accumulate;           // y += (data-gradient into an accumulating destination)
};

int mcvc_wino_input_launch(const WinoXformArgs& a, hipStream_t s);
// output transform + instance norm + activation in one launch (norm_kernels.hip); pts = 16: F(2x2,3x3), 36: F(2x2,5x5) + PixelShuffle,
// 43: F(4x4,3x3), 64: F(4x4,5x5) + PixelShuffle
struct NormArgs;
bool mcvc_norm_fwd_wino_applies(const NormArgs& a, const WinoOutArgs& w, int pts);
int mcvc_norm_fwd_wino_launch(const NormArgs& a, const WinoOutArgs& w, int pts, hipStream_t s);
// F(2x2,3x3) variants (4x4 tiles, 16 points, padding 1): the merged data-gradient of the stride-2 5x5 convs is a 3x3 conv
int mcvc_wino3_input_launch(const WinoXformArgs& a, hipStream_t s);
int mcvc_wino3_output_launch(const WinoOutArgs& a, hipStream_t s);
// forward of a stride-2 5x5 conv as a 3x3 stride-1 conv over the 4 input phases: channel k = 4*ci + 2*p + q is the plane
// x[ci][2i+p][2j+q]; a.C = 4*Cin, a.H x a.W = the phase-plane size (= conv output size), a.x_* address the ORIGINAL image
int mcvc_wino3_input_phase_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s);
// weight gradient of the stride-2 5x5 convs in the same phase formulation: tile-major operands, then
// dw[co][ci][2u'+p][2v'+q] += (G3^T dU G3)[u'][v'] for dU[16][Cout_tot][4*Cin]; output channels >= Cout go to dw1 (gate branch)
int mcvc_wino3_input_phase_t_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s);
int mcvc_wino3_dy_t_launch(const WinoXformArgs& a, hipStream_t s);
int mcvc_wino3_dw_launch(const float* du, float* dw0, float* dw1, int Cout, int nbr, int Cin, hipStream_t s);
// weight-gradient operands, tile-major (the tile index is the contraction dimension there): Vt[36][NTp][C] from x, and
// dMt[36][NTp][C] = A dY A^T from the 2x2 output-gradient tiles; rows of tiles >= NT are written as zeros
int mcvc_wino_input_t_launch(const WinoXformArgs& a, hipStream_t s);
int mcvc_wino_dy_t_launch(const WinoXformArgs& a, hipStream_t s);
// dw[co][ci][5][5] += G^T dU G  from dU[36][Cout][Cin]
int mcvc_wino_dw_launch(const float* du, float* dw, int Cout, int Cin, hipStream_t s);
int mcvc_wino_output_launch(const WinoOutArgs& a, hipStream_t s);
// Weight transform U = G g G^T, one thread per (co, ci), called from the whole-network re-pack kernel:
// U[xi][k][col] (K-major, `ld` columns, xi_stride floats between transform points) from OIHW weights w[Cout][Cin][5][5].
//   dgrad == 0: k = ci, col = co_off + co             (forward)
//   dgrad == 1: k = co_off + co, col = ci, taps flipped (data-gradient)
// G (6x5) for points {0, 1, -1, 2, -2, inf}
// (..._core: the transform of ONE filter whose 25 taps are at g -- global memory for the re-pack kernel, an LDS tile of freshly updated
//  weights for the fused optimizer step, pack_kernels.hip)
static __device__ __forceinline__ void wino_weight_core(const float* g, float* dst, int co, int ci, int ld, long long xi_stride, int co_off, int dgrad)
{
    float gg[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int l = 0; l < 5; ++l) gg[k][l] = dgrad ? g[(4 - k) * 5 + (4 - l)] : g[k * 5 + l];
    auto grow = [](const float v[5], float o[6]) {
        const float s_even = v[0] + v[2] + v[4], s_odd = v[1] + v[3];
        o[0] = 0.25f * v[0];
        o[1] = -(s_even + s_odd) * (1.0f / 6.0f);
        o[2] = -(s_even - s_odd) * (1.0f / 6.0f);
        const float e2 = v[0] * (1.0f / 24.0f) + v[2] * (1.0f / 6.0f) + v[4] * (2.0f / 3.0f);
        const float o2 = v[1] * (1.0f / 12.0f) + v[3] * (1.0f / 3.0f);
        o[3] = e2 + o2;
        o[4] = e2 - o2;
        o[5] = v[4];
    };
    float t[5][6];                                  // t[k][b] = sum_l g[k][l] G[b][l]
#pragma unroll
    for (int k = 0; k < 5; ++k) grow(gg[k], t[k]);
    const long long row = dgrad ? (co_off + co) : ci;
    const long long col = dgrad ? ci : (co_off + co);
    float* d0 = dst + row * ld + col;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float colv[5], o[6];
#pragma unroll
        for (int k = 0; k < 5; ++k) colv[k] = t[k][b];
        grow(colv, o);                              // U[a][b] = sum_k G[a][k] t[k][b]
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) d0[(long long)(aa * 6 + b) * xi_stride] = o[aa];
    }
}
static __device__ __forceinline__ void wino_weight_tile(const float* w, float* dst, int Cout, int Cin, int ld, long long xi_stride, int co_off, int dgrad, int bx, int by)
{
    // forward: threads along co (columns of the K-major matrix); data-gradient: threads along ci
    const int a_idx = bx * 256 + threadIdx.x;       // fastest index: co (forward) / ci (dgrad)
    const int b_idx = by;                           //                ci (forward) / co (dgrad)
    const int co = dgrad ? b_idx : a_idx, ci = dgrad ? a_idx : b_idx;
    if (co >= Cout || ci >= Cin) return;
    wino_weight_core(w + ((long long)co * Cin + ci) * 25, dst, co, ci, ld, xi_stride, co_off, dgrad);
}


// Batched product of the 36 Winograd points:  C[xi][m][n] = sum_k A[xi][k][m] * B[xi][k][n]   (all operands K-major)
struct WinoGemmArgs {
    const float* a; long long a_xi; int lda;     // U: [36][K (+pad row)][lda]
    const float* b; long long b_xi; int ldb;     // V: [36][K][ldb]
    float* c; long long c_xi; int ldc;           // M: [36][M][ldc]
    int M, N, K;                                 // N = valid columns (multiple of 32)
    int nxi;                                     // transform points: 36 (F(2x2,5x5)) or 16 (F(2x2,3x3)); 0 = 36
    int nt, mt;                                  // (filled by the launcher) tiles along n and m
    int mgroup;                                  // (filled by the launcher) gemm2_kernel: row tiles per group of its tile order; 0 = column tile fastest
};
int mcvc_wino_gemm_launch(const WinoGemmArgs& a, hipStream_t s);

// G of the 3-tap transforms: P = 4: F(2x2,3x3), points {0, 1, -1, inf};  P = 6: F(4x4,3x3), points {0, 1, -1, 2, -2, inf}
template <int P>
static __device__ __forceinline__ void wino_g3(float v0, float v1, float v2, float* o)
{
    if constexpr (P == 4) { o[0] = v0; o[1] = 0.5f * (v0 + v1 + v2); o[2] = 0.5f * (v0 - v1 + v2); o[3] = v2; }
    else {
        const float e = v0 + v2, e2 = v0 * (1.0f / 24.0f) + v2 * (1.0f / 6.0f);
        o[0] = 0.25f * v0; o[1] = -(e + v1) * (1.0f / 6.0f); o[2] = -(e - v1) * (1.0f / 6.0f);
        o[3] = e2 + v1 * (1.0f / 12.0f); o[4] = e2 - v1 * (1.0f / 12.0f); o[5] = v2;
    }
}

// Weight transform for the F(2x2,3x3) data-gradient of a stride-2 5x5 conv (padding 2).  The data-gradient is ONE stride-1
// 3x3 conv over dY with 4*Cin output channels (column 4*ci + 2*qh + qw = output parity class), whose taps are
//   g'[u'][v'] = W[co][ci][kh(u',qh)][kw(v',qw)],  kh(u',0) = 4 - 2u',  kh(u',1) = 5 - 2u' (u' >= 1, else no tap)
// (pack_dgrad_tile builds the same matrix for the direct kernel).  U = G g' G^T with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// One thread per (co, ci), ci fastest: stores 4 consecutive columns x 16 points.
template <int P>
static __device__ __forceinline__ void wino3_weight_core_p(const float* g, float* dst, int co, int ci, int ld, long long xi_stride, int co_off)
{
    float gg[5][5];
#pragma unroll
    for (int k = 0; k < 25; ++k) gg[k / 5][k % 5] = g[k];
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
        for (int qw = 0; qw < 2; ++qw) {
            float t[3][P];                       // t[u'][b] = sum_v' g'[u'][v'] G[b][v']
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int kh = qh ? 5 - 2 * u : 4 - 2 * u;          // qh = 1, u = 0 -> 5: no such tap
                float r0 = 0.f, r1 = 0.f, r2 = 0.f;
                if (kh <= 4) {
                    if (qw) { r1 = gg[kh][3]; r2 = gg[kh][1]; }
                    else { r0 = gg[kh][4]; r1 = gg[kh][2]; r2 = gg[kh][0]; }
                }
                wino_g3<P>(r0, r1, r2, t[u]);
            }
            float* d0 = dst + (long long)(co_off + co) * ld + 4 * ci + 2 * qh + qw;
#pragma unroll
            for (int b = 0; b < P; ++b) {
                float o[P];
                wino_g3<P>(t[0][b], t[1][b], t[2][b], o);            // U[a][b] = sum_u' G[a][u'] t[u'][b]
#pragma unroll
                for (int aa = 0; aa < P; ++aa) d0[(long long)(aa * P + b) * xi_stride] = o[aa];
            }
        }
}

template <int P>
static __device__ __forceinline__ void wino3_weight_tile_p(const float* w, float* dst, int Cout, int Cin, int ld, long long xi_stride, int co_off, int bx, int by)
{
    const int ci = bx * 256 + threadIdx.x, co = by;
    if (co >= Cout || ci >= Cin) return;
    wino3_weight_core_p<P>(w + ((long long)co * Cin + ci) * 25, dst, co, ci, ld, xi_stride, co_off);
}

// Forward twin: y[oh][ow] = sum w[kh][kw] x[2oh+kh-2][2ow+kw-2] = 3x3 stride-1 pad-1 correlation over the phase planes
// X[4ci+2p+q][i][j] = x[ci][2i+p][2j+q] with taps g'[u'][v'] = w[co][ci][2u'+p][2v'+q] (absent when the index exceeds 4).
// U[xi][k = 4ci+2p+q][col = co_off + co].  One thread per (co, ci), co fastest.
template <int P>
static __device__ __forceinline__ void wino3_weight_fwd_core_p(const float* g, float* dst, int co, int ci, int ld, long long xi_stride, int co_off)
{
    float gg[5][5];
#pragma unroll
    for (int k = 0; k < 25; ++k) gg[k / 5][k % 5] = g[k];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float t[3][P];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int kh = 2 * u + p;
                float r0 = 0.f, r1 = 0.f, r2 = 0.f;
                if (kh <= 4) { r0 = gg[kh][q]; r1 = gg[kh][2 + q]; r2 = q ? 0.f : gg[kh][4]; }
                wino_g3<P>(r0, r1, r2, t[u]);
            }
            float* d0 = dst + (long long)(4 * ci + 2 * p + q) * ld + co_off + co;
#pragma unroll
            for (int b = 0; b < P; ++b) {
                float o[P];
                wino_g3<P>(t[0][b], t[1][b], t[2][b], o);
#pragma unroll
                for (int aa = 0; aa < P; ++aa) d0[(long long)(aa * P + b) * xi_stride] = o[aa];
            }
        }
}

template <int P>
static __device__ __forceinline__ void wino3_weight_fwd_tile_p(const float* w, float* dst, int Cout, int Cin, int ld, long long xi_stride, int co_off, int bx, int by)
{
    const int co = bx * 256 + threadIdx.x, ci = by;
    if (co >= Cout || ci >= Cin) return;
    wino3_weight_fwd_core_p<P>(w + ((long long)co * Cin + ci) * 25, dst, co, ci, ld, xi_stride, co_off);
}

static __device__ __forceinline__ void wino3_weight_tile(const float* w, float* dst, int Cout, int Cin, int ld, long long xi_stride, int co_off, int bx, int by)
{ wino3_weight_tile_p<4>(w, dst, Cout, Cin, ld, xi_stride, co_off, bx, by); }
static __device__ __forceinline__ void wino3_weight_fwd_tile(const float* w, float* dst, int Cout, int Cin, int ld, long long xi_stride, int co_off, int bx, int by)
{ wino3_weight_fwd_tile_p<4>(w, dst, Cout, Cin, ld, xi_stride, co_off, bx, by); }

// ---- F(4x4,3x3) twins of the wino3 family (wino43_kernels.hip): 6x6 windows at stride 4, 36 points, 4x4 outputs per tile; the same operand
// layouts with 36 matrices (weights: wino3_weight_tile_p<6> / wino3_weight_fwd_tile_p<6>)
int mcvc_wino43_input_launch(const WinoXformArgs& a, hipStream_t s);
int mcvc_wino43_input_phase_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s);
int mcvc_wino43_output_launch(const WinoOutArgs& a, hipStream_t s);
int mcvc_wino43_input_phase_t_launch(const WinoXformArgs& a, int XH, int XW, hipStream_t s);
int mcvc_wino43_dy_t_launch(const WinoXformArgs& a, hipStream_t s);
int mcvc_wino43_dw_launch(const float* du, float* dw0, float* dw1, int Cout, int nbr, int Cin, hipStream_t s);
