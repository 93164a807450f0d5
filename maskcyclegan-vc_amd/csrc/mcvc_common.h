// Shared declarations for the MaskCycleGAN-VC gfx950 (MI355X / CDNA4) kernel library.
// Everything here is internal; the exported C ABI is include/mcvc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MCVC_OK 0
#define MCVC_ERR_INVALID 1001
#define MCVC_ERR_WORKSPACE 1002

// Process-wide switch (mcvc_set_deterministic / env MCVC_DETERMINISTIC=1): every accumulation that would use floating-point
// atomics (order depends on workgroup scheduling) takes a fixed-order path instead -- private slabs summed by the consumer.
// The reference's CPU path is bit-reproducible run to run; this mode restores that property.
int mcvc_deterministic();

// ------------------------------------------------------------------------------------------------
// Planner knobs.  In the product every knob IS its default: the call below is a constant, nothing reads the environment (the one
// environment switch of the library is MCVC_DETERMINISTIC, net.hip).  The tuning tools (tools/conv_tune.py, wgrad_tune.py, gemm_*.py,
// ab_*.sh) run against an EXPERIMENTS build of the same sources -- `MCVC_EXPERIMENTS=1 python __graft_entry__.py` compiles them with
// -DMCVC_EXPERIMENTS into lib/libmcvc_hip_exp.so, selected with MCVC_LIB -- in which MCVC_<NAME> overrides the default, so every choice
// recorded in DESIGN.md can be re-measured without shipping its switch.
// ------------------------------------------------------------------------------------------------
#ifdef MCVC_EXPERIMENTS
#include <cstdlib>
static inline int mcvc_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool mcvc_knob_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline int mcvc_knob(const char*, int dflt) { return dflt; }
static inline bool mcvc_knob_set(const char*) { return false; }
#endif

static inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }
static inline long long cdiv_ll(long long a, long long b) { return (a + b - 1) / b; }
static inline int round_up_i(int a, int b) { return cdiv_i(a, b) * b; }

// ------------------------------------------------------------------------------------------------
// Direct convolution (forward and data-gradient share one kernel; the data-gradient is a
// convolution of dY with re-packed weights).  Implicit GEMM on v_mfma_f32_32x32x2_f32:
//   M = output channel (co), N = output pixel, K = (input channel, kh, kw).
// The input patch of a pixel tile is staged ONCE in LDS per channel chunk and every tap reads a
// shifted window of it -- no im2col buffer exists anywhere in the library (r5: the GEMM-shaped layers -- the discriminators'
// 3 x 3 stride-2 convolutions, the 1-D trunk beyond the fused kernels -- gather their operands in place too: sgemm.h).
// ------------------------------------------------------------------------------------------------
enum ConvOutMode { CONV_OUT_SLAB = 0, CONV_OUT_ACCUM = 1 };

struct ConvArgs {
    const float* x;        // input, element (n, c, h, w) at x + n*x_sb + c*x_sc + h*x_sh + w
    const float* w;        // packed weights [w_rows = Cin_pad*KH*KW][w_cout], row = (ci*KH+kh)*KW+kw
    const float* bias;     // [Cout] or nullptr
    float* y;              // split 0 destination
    float* y_slabs;        // destination of split s>=1: y_slabs + (s-1)*slab_stride
    long long x_sb;
    long long x_sc;
    long long y_sb;
    long long y_sc;
    long long slab_stride;
    int x_sh;
    int y_sh, y_sw;
    int Cin, H, W;         // logical input dims
    int Cout, OH, OW;      // logical output grid
    int KH, KW, stride, pad_h, pad_w;
    int w_rows, w_cout;
    int cic;               // input channels per LDS chunk (even)
    int nchunks, chunks_per_split, nsplit;
    int tow_log2;          // pixel tile width = 1 << tow_log2 (8, 16 or 32)
    int tiles_w, tiles_total, nb;
    int PH, PW, PWp, PWh, plane, xs_floats;   // LDS patch geometry
    int out_mode;          // ConvOutMode
    int shuffle;           // 1: PixelShuffle(2) store  y[c=co>>2][2oh+((co>>1)&1)][2ow+(co&1)]
    int YH, YW;            // shuffle mode: bounds of the shuffled image (rows/cols beyond are not stored)
    long long w_nstride;   // image n uses the packed weights at w + n*w_nstride (0: shared; Winograd: one matrix per transform point)
    int gemm;              // 1: GEMM mode (1x1 conv whose pixel tiles are contiguous): the input patch is streamed by LDS-DMA too
};

struct ConvProblem {
    int Cin, H, W;         // input
    int Cout, OH, OW;      // output grid
    int KH, KW, stride, pad_h, pad_w;
};

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[co][ci][kh][kw] += sum_{n,oh,ow} dY[n][co][oh][ow] * X[n][ci][oh*s+kh-p][ow*s+kw-p]
//   M = co, N = ci (or (ci,kw) for tiny Cin), K = pixel; one accumulator per tap.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dy;
    float* dw;
    float* slabs;          // ksplit private partial-dW slabs (ksplit > 1)
    long long dw_floats, slab_stride;
    long long x_sb, x_sc;
    long long dy_sb, dy_sc;
    int x_sh, dy_sh;
    int NB, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad_h, pad_w;
    int toh, tow, tiles_h, tiles_w;
    int PH, PW, PWp, plane;
    int pitch_a;
    int ksplit, ci_tiles;
    int lane_mode;         // 0: N lane = ci ; 1: N lane = ci*KW + kw  (Cin*KW <= 32)
    int nkwg;              // kw groups per kh (lane_mode 0)
    int cot;               // output channels per block
    int atomic;            // K-split workgroups add into dw with atomics (tiny dW); otherwise private slabs + reduce
};

// ------------------------------------------------------------------------------------------------
// InstanceNorm (+ activation) forward / backward
// ------------------------------------------------------------------------------------------------
enum ActKind { ACT_NONE = 0, ACT_GLU = 1, ACT_SILU = 2, ACT_SIGMOID = 3 };

struct NormArgs {
    // conv output (pre-norm), dense planes: plane (n,cx) at x + n*x_sn + cx*x_sc, P = H*W contiguous floats.
    // For ACT_GLU the value plane is channel c and the gate plane channel C + c.
    float* x;                 // slab 0 (also receives the slab-reduced sum)
    const float* x_slabs;     // slabs 1..nslab-1 (same layout), slab_stride floats apart
    long long x_sn, x_sc, slab_stride;
    int nslab;
    const float* gamma[2];    // [C] affine weight of the value / gate branch
    const float* beta[2];
    float* stats;             // [N][Cx][2] = mean, rstd  (Cx = C or 2C)
    float* y;                 // output, plane (n,c) at y + n*y_sn + c*y_sc, element (h,w) at + h*y_sh + w
    const float* res;         // optional residual, same addressing as y
    long long y_sn, y_sc;
    int y_sh;
    int N, C, H, W;
    int act;
    float eps;
    // y in the phase-split padded layout of sgemm.h ("xs": the next layer is a 3x3 stride-2 convolution run as an implicit GEMM): plane (n, c)
    // at y + n*y_sn + c*y_sc holds four sub-planes of xs_plane floats, rows of xs_pw floats; the kernel also writes the zero borders
    int y_xs, xs_pw; long long xs_plane;
};

struct NormBwdArgs {
    const float* x;           // reduced conv output (dense planes)
    long long x_sn, x_sc;
    const float* gamma[2];
    const float* beta[2];
    const float* stats;
    float* dy;                // grad wrt y (slab 0; receives slab-reduced sum when nslab>1), addressed like y
    const float* dy_slabs;
    long long y_sn, y_sc, slab_stride;
    int y_sh;
    int nslab;
    float* dx;                // grad wrt conv output: plane (n,cx) at dx + n*dx_sn + cx*dx_sc (dense) unless unshuffle
    long long dx_sn, dx_sc;
    int dx_sh;
    int unshuffle;            // 1: plane (n,c) element (h,w) -> conv channel 4c+2(h&1)+(w&1), pixel (h>>1, w>>1), row pitch dx_sh
    int dx_pitch;             // > W (dense mode only): rows of dx_pitch floats, H + 1 rows per plane; the kernel writes zeros beyond column W - 1 and
                              // in row H (the implicit-GEMM data gradient reads one column / row past the image: sgemm.h)
    float* dgamma[2];         // accumulated (+=); may be null
    float* dbeta[2];
    int N, C, H, W;
    int act;
};

// element-wise activation without a norm (conv1 GLU, D convLayer1 SiLU, D output sigmoid, plain slab reduce)
struct ActArgs {
    float* x;                 // [N][Cx][P] dense; slab 0, receives reduced sum
    const float* x_slabs;
    long long slab_stride;
    int nslab;
    float* y;                 // [N][C][P] dense
    int N, C, P;
    int act;
};

struct ActBwdArgs {
    const float* x;           // reduced pre-activation [N][Cx][P]
    float* dy;                // [N][C][P] slab 0
    const float* dy_slabs;
    long long slab_stride;
    int nslab;
    float* dx;                // [N][Cx][P]
    int N, C, P;
    int act;
};

// ------------------------------------------------------------------------------------------------
// host-side launch helpers (defined in the .hip files)
// ------------------------------------------------------------------------------------------------
struct ConvIO {
    const float* x; long long x_sb, x_sc; int x_sh;
    float* y; long long y_sb, y_sc; int y_sh, y_sw;
    float* slabs; long long slab_stride;                  // scratch for split-K partial slabs (s >= 1)
    int nsplit;                                            // exact K-split count (>=1); planned by the caller
    int accumulate;                                        // 1: y += conv (atomic when nsplit > 1)
    int shuffle;
    int YH, YW;                                            // shuffle bounds (0 = 2*OH, 2*OW)
    long long w_nstride;                                   // per-image weight stride in floats (0 = all images share w)
    int gemm_ok;                                           // caller guarantees contiguous pixel rows (x_sh == W): GEMM mode may be used
    int tile_cfg;                                          // 0 = planner's choice; k > 0 forces tile configuration k-1 (0 = 128x128, 1 = 128x64, ...)
};

int mcvc_conv_launch(const ConvProblem& p, int NB, const ConvIO& io, const float* wpk, int w_rows, int w_cout,
                     const float* bias, hipStream_t s, int* nsplit_out);
// number of split-K slabs the planner would use (for workspace sizing)
int mcvc_conv_plan_nsplit(const ConvProblem& p, int NB, int allow_split);
// few-output-channel VALU path (fewout_kernels.hip); mcvc_conv_plan_nsplit / mcvc_conv_launch route to it when it applies
bool mcvc_fewout_applies(const ConvProblem& p);
int mcvc_fewout_plan_nsplit(const ConvProblem& p, int NB, int allow_split);
int mcvc_fewout_launch(const ConvProblem& p, int NB, const ConvIO& io, const float* wpk, int w_cout, const float* bias, hipStream_t s);
// the discriminators' output layer (1x3, C -> 1, + sigmoid) and its data-gradient; w / bias = the parameters themselves
int mcvc_disc_out_fwd_launch(const float* x, const float* w, const float* bias, float* logit, float* out, int NB, int C, int H, int W, hipStream_t s);
// the discriminators' first layer (3x3 from one channel) + x*sigmoid(x): c0 = pre-activation, y0 = activation, dense [NB][Cout][H][W]
// xs != 0: y0 in the phase-split padded layout of sgemm.h (borders zeroed)
int mcvc_disc_conv1_fwd_launch(const float* x, const float* w, const float* bias, float* c0, float* y0, int NB, int Cout, int H, int W, hipStream_t s, int xs = 0);
int mcvc_disc_out_dgrad_launch(const float* dlogit, const float* w, float* dx, int NB, int C, int H, int W, hipStream_t s);

struct WgradIO {
    const float* x; long long x_sb, x_sc; int x_sh;
    const float* dy; long long dy_sb, dy_sc; int dy_sh;
};
int mcvc_wgrad_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, float* slabs, long long slab_cap_floats, hipStream_t s);
// conv1's weight gradient (Cin <= 2, 5 x 15) on the matrix cores (fewout_kernels.hip); dw accumulates
bool mcvc_wgrad_cin2_applies(const ConvProblem& p, const WgradIO& io);
int mcvc_wgrad_cin2_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, hipStream_t s);
// one-output-channel weight gradient on the VALU (fewout_kernels.hip): lastConvLayer, the discriminator's output conv
bool mcvc_wgrad_cout1_applies(const ConvProblem& p);
// one-INPUT-channel 3x3 weight gradient (the discriminators' first conv)
bool mcvc_wgrad_cin1_applies(const ConvProblem& p);
int mcvc_wgrad_cin1_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, hipStream_t s);
int mcvc_wgrad_cout1_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, hipStream_t s);
// scratch floats the K-split of this weight gradient wants (0 = no split)
long long mcvc_wgrad_plan_slab_floats(const ConvProblem& p, int NB);

// Batched small-K weight gradients (1-D trunk at small batch): all of a backward pass's trunk layers in ONE launch.
// Every job is dW[co][ci][kw] += sum_{b,t} dY[co][b][t] * X[ci][b][t + kw - 1]  with dY, X in trunk layout [C][B][T4].
struct SmallKJob { const float* x; const float* dy; float* dw; int Cin, Cout; int xB; };   // xB: samples per channel of x (0 = B)
#define MCVC_SMALLK_MAX_JOBS 24
int mcvc_wgrad_smallk_batch_launch(const SmallKJob* jobs, int njobs, int B, int T4, hipStream_t s);
bool mcvc_wgrad_smallk_batch_applies(int B, int T4);

int mcvc_norm_fwd_launch(const NormArgs& a, hipStream_t s);
int mcvc_norm_bwd_launch(const NormBwdArgs& a, hipStream_t s);
int mcvc_act_fwd_launch(const ActArgs& a, hipStream_t s);
int mcvc_act_bwd_launch(const ActBwdArgs& a, hipStream_t s);
