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
};

bool mcvc_trunk_applies(int Cin, int KW, int M, int B, int T4, int mode, int ksplit);
long long mcvc_trunk_lds_floats(int Cin, int KW, int B, int T4, int ksplit);
int mcvc_trunk_launch(const TrunkArgs& a, int ksplit, hipStream_t s);
int mcvc_fill_rows_launch(float* dst, const float* bias, int C, int per_row, hipStream_t s);
