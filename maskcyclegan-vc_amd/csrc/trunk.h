// Fused small-batch 1-D trunk layer (conv1d + bias + InstanceNorm + GLU / residual) and its transposed weight pack.
#pragma once
#include <hip/hip_runtime.h>

enum TrunkMode { TRUNK_PLAIN = 0, TRUNK_IN = 1, TRUNK_IN_GLU = 2 };

struct TrunkArgs {
    const float* a0; const float* a1;       // row-major [M][K] weights (OIHW conv1d tensor); a1 = gate branch (GLU)
    const float* bias0; const float* bias1;
    const float* x; long long x_sc; long long x_sb;   // input [Cin][B][T4]: x + ci*x_sc + b*x_sb + t
    int Cin, KW, K;                         // K = Cin*KW
    int M, Mtot;                            // output channels per branch / in conv_out (GLU: 2*M)
    int B, T4, N;                           // N = B*T4 <= 32
    float* conv_out; long long c_sc; long long c_sb;  // pre-norm conv output (mode 0: the result), [Mtot][B][T4]
    int accumulate;                         // mode 0 only
    int slab_all;                           // mode 0, K split: EVERY split s writes slabs + s*slab_stride and the destination is left alone
                                            // (deterministic accumulate: the consumer sums destination + all slabs in a fixed order)
    float* slabs; long long slab_stride;    // mode 0, K split without accumulate: split s >= 1 writes slabs + (s-1)*slab_stride (the consumer sums)
    float* stats;                           // [B][Mtot][2] (mean, rstd)
    const float* gamma0; const float* beta0; const float* gamma1; const float* beta1;
    float* y; const float* res; long long y_sn; long long y_sc;   // plane (b, c) at y + b*y_sn + c*y_sc
    float eps;
    int mode;
    // ---- fused InstanceNorm backward in front of a data-gradient (mode TRUNK_PLAIN): the staged input is not `x` itself but
    //      X' = IN-backward(x) of the layer being back-propagated -- every workgroup recomputes the rows of its K slice (a few hundred
    //      flops per row) instead of waiting for a separate norm_bwd launch; the workgroups with blockIdx.x == 0 also store X' (the
    //      weight-gradient kernel reads it) and accumulate d(gamma), d(beta).
    int pre;                                // 0 none; 1 plain IN (x = upstream gradient of the norm output); 2 IN + gated GLU
    int pre_C;                              // normalised channels per branch (X' has pre_C or 2*pre_C channels)
    const float* pre_x;                     // conv output of the forward pass (pre-norm), [Cx][pre_xB][T4]
    int pre_xB;                             // samples per channel of pre_x (0 = B; > B: backward over the first B samples of a larger forward pass)
    const float* pre_stats;                 // [B][Cx][2] mean, rstd
    const float* pre_gamma0; const float* pre_beta0; const float* pre_gamma1; const float* pre_beta1;
    float* pre_out;                         // X' [Cx][B][T4]
    float* pre_dgamma0; float* pre_dbeta0; float* pre_dgamma1; float* pre_dbeta1;     // accumulated (+=); nullable
};

bool mcvc_trunk_applies(int Cin, int KW, int M, int B, int T4, int mode, int ksplit);
long long mcvc_trunk_lds_floats(int Cin, int KW, int B, int T4, int ksplit);
int mcvc_trunk_launch(const TrunkArgs& a, int ksplit, hipStream_t s);
int mcvc_fill_rows_launch(float* dst, const float* bias, int C, int per_row, hipStream_t s);

// ---- persistent trunk forward: the six residual blocks + conv1dto2d (13 dependent layers) in ONE launch ---------------------------
// Layer l+1 needs every output row of layer l, so a plain launch chain is 13 dependent kernel boundaries of ~10 us each for ~1.5 us of
// arithmetic.  Here a fixed set of workgroups walks the layers; between layers they meet at an in-kernel arrival counter
// (write-through stores of the activations, one agent-scope acquire per workgroup -- cdna_hip_programming.md section 6, Guideline 16).
struct TrunkLayerDesc {
    const float* a0; const float* a1;        // OIHW weights [M][Cin*KW]; a1 = gate branch (mode TRUNK_IN_GLU)
    const float* bias0; const float* bias1;
    const float* gamma0; const float* beta0; const float* gamma1; const float* beta1;
    const float* x;                          // input [Cin][B][T4]
    float* conv_out;                         // pre-norm conv output [Mtot][B][T4]   (backward needs it)
    float* stats;                            // [B][Mtot][2]
    float* y; long long y_sn, y_sc;          // plane (b, c) at y + b*y_sn + c*y_sc
    const float* res;                        // residual, addressed like y
    int Cin, KW, M, mode;
    int rows;                                // output rows of one tile per branch: 8 (GLU: + the 8 gate rows), 4 or 16
};
#define MCVC_TRUNK_NET_LAYERS 13
struct TrunkFwdNetArgs {
    TrunkLayerDesc L[MCVC_TRUNK_NET_LAYERS];
    int nlayers, B, T4;
    int x_floats;                            // LDS floats reserved for the staged input (max over the layers)
    float eps;
    unsigned* sync;                          // [nlayers + 2] arrival counters + error word, zeroed by the launcher
    int fault_inject;                        // test hook: workgroup 0 skips its first arrival (set by the launcher)
};
int mcvc_trunk_set_fault_inject(int on);
// number of persistent trunk passes the caller keeps in flight at once (a grouped launch counts as two): the persistent kernels are used
// only while 64 x that many workgroups fit the device's compute units; returns the previous value
int mcvc_trunk_set_passes_in_flight(int n);
// Co-residency: the kernel's 64 workgroups wait for each other, so all 64 must be resident at once -- each takes a whole CU (up to 160 KB
// of LDS).  With P persistent passes in flight on different streams the device needs 64 * P <= 256 CUs for every pass to be guaranteed
// progress; the trainer runs at most two generator passes at a time (one grouped launch = 128 workgroups, or two lanes of 64).  Beyond that
// a pass can only be delayed, not deadlocked, as long as the over-subscribing passes are not ALL partially resident; the bounded spin in
// wait_arrivals turns even that case into a reported fault (NaN result + error word) instead of a hang.
// true when the persistent forward handles (B, T4) -- same regime as the per-layer fused kernels (N = B*T4 <= 64, LDS permitting: 48 at T4 = 16)
bool mcvc_trunk_net_applies(int B, int T4);
int mcvc_trunk_fwd_net_launch(TrunkFwdNetArgs& a, hipStream_t s);
#define MCVC_TRUNK_SYNC_WORDS 32

// ---- persistent trunk BACKWARD: the data-gradient chain through the six residual blocks (12 dependent layers) in ONE launch ---------
// Layer = InstanceNorm(+GLU) backward of the layer's input gradient, recomputed by every workgroup for all staged channels (a few hundred
// flops per row), then the transposed 1x3 convolution on the fused kernels' weight copy -- what trunk_layer_kernel<3, 1, PRE> does per
// launch -- with the hand-off of trunk_fwd_net_kernel between layers (write-through stores, one arrival counter per layer, sc1 loads).
// The channel's owner (channel mod workgroups) also stores X' = the gradient w.r.t. the conv output (the batched weight-gradient kernel
// reads it after the launch) and adds d(gamma), d(beta).  Every reduction runs in a fixed order: the result does not depend on timing.
struct TrunkBwdLayerDesc {
    const float* wt;                         // data-gradient weights [M][Cx*KW] (transposed + flipped copy, pack.h PACK_TRUNK_T)
    const float* dy;                         // gradient w.r.t. the norm (+GLU) output [C][B][T4]
    const float* px;                         // the forward pass's pre-norm conv output [Cx][pxB][T4]
    int pxB;                                 // samples per channel of px (0 = B; > B: backward over the first B samples of a larger forward pass)
    const float* stats;                      // [B][Cx][2] mean, rstd
    const float* g0; const float* b0; const float* g1; const float* b1;      // affine parameters (value | gate)
    float* xout;                             // X' [Cx][B][T4]
    float* dg0; float* db0; float* dg1; float* db1;                          // accumulated (+=); nullable
    float* out;                              // [M][B][T4]
    int pre, C, M, rows;                     // pre 1: plain IN (Cx = C), 2: IN + gated GLU (Cx = 2C)
    int flags;                               // TBWD_* bits
};
enum { TBWD_ACCUMULATE = 1,                  // out += result (the skip connection's gradient is already there)
       TBWD_DY_FRESH = 2,                    // dy was written by the previous layer of this launch: sc1 loads
       TBWD_SLAB_DY = 4, TBWD_SLAB_OUT = 8 };// the chain's entering gradient arrives as K-split slabs (TrunkBwdNetArgs::slabs): add them to
                                             // every read of dy / of the accumulated-onto out
#define MCVC_TRUNK_BWD_LAYERS 12
struct TrunkBwdNetArgs {
    TrunkBwdLayerDesc L[MCVC_TRUNK_BWD_LAYERS];
    int nlayers, B, T4, x_floats;
    const float* slabs; long long slab_stride; int nslab;      // nslab - 1 further slabs of the entering gradient, slab_stride floats apart
    unsigned* sync;                          // arrival counters (MCVC_TRUNK_SYNC_WORDS - 1 words, zeroed by the launcher)
    unsigned* err;                           // sticky error word (shared with the forward kernel: mcvc_gen_trunk_fault)
    int fault_inject;
};
bool mcvc_trunk_bwd_net_applies(int B, int T4);
int mcvc_trunk_bwd_net_launch(TrunkBwdNetArgs& a, hipStream_t s);
