// Opt-in per-launch timing: HIP events recorded on the launch stream around every kernel, with the
// launcher's own algorithmic FLOP / byte count attached.  Off by default (zero overhead: one branch).
// Used by bench.py for the `roofline` object; not part of the compute path.
#pragma once
#include <hip/hip_runtime.h>
#include "twin.h"

enum KernelKind {
    K_CONV_L = 0, K_CONV_M, K_CONV_N, K_CONV_T, K_CONV_S, K_CONV_S2, K_CONV_Q, K_CONV_FEW, K_WINO_GEMM,
    K_WGRAD_2x5, K_WGRAD_1x5, K_WGRAD_2x3, K_WGRAD_1x3, K_WGRAD_4x1, K_WGRAD_1x1, K_WGRAD_SMALLK, K_TRUNK,
    K_NORM_FWD, K_NORM_BWD, K_ACT_FWD, K_ACT_BWD, K_PACK, K_BIAS_GRAD, K_LOSS, K_ADAM, K_ELEMENTWISE, K_SGEMM,
    K_COUNT
};

extern bool g_mcvc_trace_on;
void mcvc_trace_begin_(int kind, hipStream_t s, double flops, double bytes);
void mcvc_trace_end_(hipStream_t s);

// (grouped launches, launch.h: nothing is launched while a pass is being recorded; the launching walk counts both networks' work)
struct TraceScope {
    hipStream_t s; bool on;
    TraceScope(int kind, hipStream_t st, double flops, double bytes) : s(st), on(g_mcvc_trace_on && mcvc_twin_phase() != 1)
    {
        if (on) { const double k = mcvc_twin_phase() == 2 ? 2.0 : 1.0; mcvc_trace_begin_(kind, s, k * flops, k * bytes); }
    }
    ~TraceScope() { if (on) mcvc_trace_end_(s); }
};
