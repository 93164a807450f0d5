// bf16 inference path of the Generator (BASELINE configs[4]: generator_A2B, bs=16, 80 x 512 frames, bf16):
// NHWC bf16 activations, bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation, fp32 InstanceNorm statistics.
// Kernels in bf16_kernels.hip, layer schedule + C ABI in infer_bf16.hip.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned short bf16_t;          // storage type (raw bits)

// ---- implicit-GEMM convolution, NHWC, im2col-free ----------------------------------------------------------------
//   M = output channel, N = output pixel (a TH x TW rectangle of one image), K = (kh, ci-chunk of 32, kw, ci)
// x: element (n, h, w, c) at x + n*x_sn + h*x_sh + w*x_sw + c   (channels contiguous, Cin % 32 == 0)
// w: packed [Cout_pad][KH][Cin/32][KW][32] bf16, Cout_pad % BM == 0 (extra rows zero)
// y: element (n, oh, ow, co) at y + n*y_sn + oh*y_sh + ow*y_sw + co, co < Cout
struct Bf16ConvArgs {
    const bf16_t* x; long long x_sn; int x_sh, x_sw;
    const bf16_t* w;
    const float* bias;                  // [Cout_pad] or nullptr
    bf16_t* y; long long y_sn; int y_sh, y_sw;
    int N, H, W, Cin, Cout, Cout_pad, OH, OW, KH, KW, stride, pad_h, pad_w;
    int TH, tw_log2, tiles_h, tiles_w;  // pixel tile = TH x (1 << tw_log2) = 128 or 64 pixels (chosen by the launcher)
    int PH, PW;                         // input patch of a tile: (TH-1)*stride + KH rows, (TW-1)*stride + KW columns
    int patch_bytes;                    // LDS bytes reserved for the staged patch (>= PH*PW*64, and >= prefetch depth * 4096 so that every thread stores every piece)
    int wbufs;                          // LDS weight buffers (2 = double buffer; 1 when that lets two workgroups share a CU)
    int glu;                            // 1: rows [0, Cout_pad/2) are value channels, [Cout_pad/2, Cout_pad) their gates, interleaved per
                                        //    64-row block by the packer; the epilogue stores value * sigmoid(gate): Cout = Cout_pad / 2
};
int mcvc_bf16_conv_launch(const Bf16ConvArgs& a, hipStream_t s);
// pixel-tile shape for an OH x OW output grid (TH * TW = bn)
void mcvc_bf16_conv_tile(int OH, int OW, int KH, int KW, int stride, int bn, int* TH, int* tw_log2);

// ---- InstanceNorm (+ activation) on NHWC bf16 ---------------------------------------------------------------------
// x: conv output, element (n, p = (h, w), cx) at x + n*x_sn + h*x_sh + w*x_sw + cx.
// shuffle = 1: the normalised tensor is PixelShuffle(2)(x): output channel c = cx / 4, pixel (2h + ((cx>>1)&1), 2w + (cx&1)).
// Statistics per (n, output channel) over all its elements, fp32, shifted sums (shift = the first element).
enum Bf16Act { BF16_ACT_NONE = 0, BF16_ACT_GLU = 1, BF16_ACT_SILU = 2 };
struct Bf16NormArgs {
    const bf16_t* x; long long x_sn; int x_sh, x_sw;
    int N, H, W, Cx;                    // conv-output channels (GLU: value C | gate C ; shuffle: 4 per output channel)
    int shuffle, act, has_norm;         // has_norm = 0: activation only (conv1's GLU)
    const float* gamma[2]; const float* beta[2];      // [C] value / gate branch
    float* partial;                     // [N][S][Cn][2] shifted (sum, sumsq) partials; Cn = number of normalised channels
    float* stats;                       // [N][Cn][2] mean, rstd
    int S;                              // pixel splits of the statistics pass
    const bf16_t* res;                  // optional residual, addressed like y
    bf16_t* y; long long y_sn; int y_sh, y_sw;        // output element (n, h', w', c) at y + n*y_sn + h'*y_sh + w'*y_sw + (c / y_csplit)*y_sc2 + c % y_csplit
    int y_csplit, y_sc2;                // channel split of the output address (0 = none)
    float eps;
};
int mcvc_bf16_norm_launch(const Bf16NormArgs& a, hipStream_t s);
long long mcvc_bf16_norm_partial_floats(const Bf16NormArgs& a);
int mcvc_bf16_norm_splits(int N, int P, int Cn);

// ---- edges -----------------------------------------------------------------------------------------------------------
// xin[b][h][w][kw*2 + ci] = (ci == 0 ? x*mask : mask)[b][h][w + kw - 7]  (kw < 15; zero outside the image; channels 30, 31 zero)
int mcvc_bf16_prep_launch(const float* x, const float* mask, bf16_t* xin, int B, int H, int W, hipStream_t s);
// conv1 + its input preparation + its gated GLU in one launch (model.py:241-242): y[b][h][w][128] bf16 from x, mask fp32 [B][H][W] (mask may
// be null = ones); w / bias = the FOLD_KW pack of conv1 | conv1_gates with glu_interleave ([256][5][32] bf16, [256] fp32)
int mcvc_bf16_conv1_fused_launch(const float* x, const float* mask, const bf16_t* w, const float* bias, bf16_t* y, int B, int H, int W, hipStream_t s);
// out[b][h][w] = bias + sum_kw z[b][h][w + kw - 7][kw]      (z: [B][H][W][32] bf16, kw < 15) -- the last 5x15 conv's kw reduction
int mcvc_bf16_last_launch(const bf16_t* z, const float* bias, float* out, int B, int H, int W, hipStream_t s);

// the last conv (128 -> 1, 5 x 15) + its kernel-column sum + bias in one launch: out[b][h][w] fp32 from x [B][H][W][128] bf16; w = the KW_OUT
// pack ([32][5][4][32] bf16: row = kernel column), bias = the layer's scalar bias (device pointer) or null
int mcvc_bf16_last_fused_launch(const bf16_t* x, const bf16_t* w, const float* bias, float* out, int B, int H, int W, hipStream_t s);

// ---- a residual-block layer in one launch (r6): y = {IN(conv1d_3(x; w0)) * sigmoid(IN(conv1d_3(x; w1))) | IN(conv1d_3(x; w0)) + res} on [B][W][C] bf16 rows,
// W <= 128 (T <= 512 frames), Cin in {256, 512}, C % 32 == 0.  w: mcvc_bf16_trunk_pack_launch's operand-order copy (elems() bf16).
bool mcvc_bf16_trunk_layer_applies(int W, int Cin, int C);
long long mcvc_bf16_trunk_pack_elems(int Cin, int C, int glu);
int mcvc_bf16_trunk_pack_launch(const float* w0, const float* w1, bf16_t* dst, int Cin, int C, int glu, hipStream_t s);
int mcvc_bf16_trunk_layer_launch(const bf16_t* x, long long x_sn, const bf16_t* w, const float* g0, const float* b0, const float* g1, const float* b1,
                                 const bf16_t* res, bf16_t* y, long long y_sn, int B, int W, int Cin, int C, int glu, float eps, hipStream_t s);

// ---- conv2dto1d (5120 -> 256, k = 1) + its InstanceNorm in one launch (r6): x [B][W][5120] bf16 (channel h * 256 + c), y [B][W][256]; W <= 128
bool mcvc_bf16_c2d1d_applies(int W);
long long mcvc_bf16_c2d1d_pack_elems(void);
int mcvc_bf16_c2d1d_pack_launch(const float* w, bf16_t* dst, hipStream_t s);
int mcvc_bf16_c2d1d_launch(const bf16_t* x, long long x_sn, const bf16_t* w, const float* gamma, const float* beta, bf16_t* y, long long y_sn,
                           int B, int W, float eps, hipStream_t s);

// ---- weight packing (fp32 OIHW parameters -> bf16 [Cout_pad][KH][Cin/32][KW][32]) -------------------------------------
enum Bf16PackKind {
    BF16_PACK_PLAIN = 0,       // k index = ci (Cin_src % 32 == 0)
    BF16_PACK_FOLD_KW = 1,     // conv1: packed KW = 1, channel kw*Cin_src + ci  (Cin_src*KW_src <= 32)
    BF16_PACK_HC_IN = 2,       // conv2dto1d: packed channel h*256 + c  <- source channel c*20 + h
    BF16_PACK_HC_OUT = 3,      // conv1dto2d: packed row h*256 + c      <- source row c*20 + h
    BF16_PACK_KW_OUT = 4,      // last conv: packed row kw (15 of 32), KW = 1  <- source [0][ci][kh][kw]
};
struct Bf16PackArgs {
    const float* w[2];          // source OIHW tensors of up to two branches (value | gate), [Cout_src][Cin_src][KH][KW_src]
    bf16_t* dst;
    int kind, nbr, Cout_src, Cin_src, KH, KW_src;
    int Cout_pad, KW, Cin;      // packed geometry
    int glu_interleave;         // 1: 64-row blocks = [32 value rows | 32 gate rows] (fused-GLU epilogue)
};
int mcvc_bf16_pack_launch(const Bf16PackArgs& a, hipStream_t s);
// fp32 vectors (bias / gamma / beta): kind PLAIN: dst = [src | src2] (src2 optional, n_src each); HC_OUT: dst[h*256+c] = src[c*20+h];
// kind -1: 64-element blocks [32 of src | 32 of src2] (fused-GLU row order).  Elements beyond the sources are zero.
int mcvc_bf16_vec_launch(const float* src, const float* src2, float* dst, int n_src, int n_dst, int kind, hipStream_t s);
