// InstanceNorm(affine, eps=1e-5, biased variance) fused with the activation that follows it in the
// reference, forward and backward, plus the norm-less activations.
//
// Reference call sites (mask_cyclegan_vc/model.py): nn.InstanceNorm{1,2}d + gated GLU
// `a * sigmoid(g)` (:71-76, :101-103), + `x*sigmoid(x)` (:20-21 used at :236, :337), plain IN
// (:147-148, :188-189, :68-69 followed by the residual add of :76).
//
// A group of G lanes (16 / 64 / 256) owns one (n, c) plane; reductions are wavefront shuffles
// (plus one LDS hop for G = 256).  Statistics are two-pass (mean, then centred second moment) in
// fp32 -- planes are as small as 16 elements in the 1-D trunk, where a one-pass E[x^2]-E[x]^2
// would lose the parity budget.  Split-K partial slabs written by the conv kernel are summed here
// (the "launch-boundary reduce"), so the conv never needs atomics for its K split.
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include "wino.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int G>
__device__ __forceinline__ float gsum(float v, float* red)
{
    if constexpr (G <= 64) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// zero borders of one (n, c) plane group in the phase-split padded layout (row 0 and columns 0..3 of the four sub-planes), by a group of G lanes
template <int G>
__device__ __forceinline__ void xs_zero_borders(float* y0, int H, int pw, long long plane, int l)
{
    const int per = pw + 4 * (H >> 1);
    for (int i = l; i < 4 * per; i += G) {
        const int pq = i / per, j = i - pq * per;
        float* p = y0 + pq * plane;
        if (j < pw) p[j] = 0.f;
        else { const int r = (j - pw) >> 2, cc = (j - pw) & 3; p[(long long)(r + 1) * pw + cc] = 0.f; }
    }
}
__device__ __forceinline__ long long xs_off(int h, int w, int pw, long long plane)
{
    return (long long)((h & 1) * 2 + (w & 1)) * plane + (long long)((h >> 1) + 1) * pw + (w >> 1) + 4;
}
// zero borders of one padded dY plane (columns W .. pitch-1 of rows 0 .. H-1, and row H)
template <int G>
__device__ __forceinline__ void dyp_zero_borders(float* d0, int H, int W, int pitch, int l)
{
    const int pc = pitch - W, per = H * pc + pitch;
    for (int i = l; i < per; i += G) {
        if (i < H * pc) { const int r = i / pc, cc = i - r * pc; d0[(long long)r * pitch + W + cc] = 0.f; }
        else d0[(long long)H * pitch + (i - H * pc)] = 0.f;
    }
}

template <int G>
__global__ void __launch_bounds__(256) norm_fwd_kernel(const Twin<NormArgs> tw)
{
    const NormArgs a = tw.v[blockIdx.z];
    __shared__ float red[4];
    constexpr int GPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const long long plane_id = (long long)blockIdx.x * GPB + g;
    if (plane_id >= (long long)a.N * a.C) return;      // whole groups leave together (G<=64); G=256: never taken
    const int n = (int)(plane_id / a.C), c = (int)(plane_id - (long long)n * a.C);
    const int P = a.H * a.W;
    const float invP = 1.0f / (float)P;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    const int Cx = a.C * nbr;
    float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
    float* xp[2];
    for (int br = 0; br < nbr; ++br) {
        const int cx = c + br * a.C;
        const long long poff = (long long)n * a.x_sn + (long long)cx * a.x_sc;
        float* x = a.x + poff;
        xp[br] = x;
        float s = 0.f;
        for (int i = l; i < P; i += G) {
            float v = x[i];
            if (a.nslab > 1) {
                for (int sl = 1; sl < a.nslab; ++sl) v += a.x_slabs[(long long)(sl - 1) * a.slab_stride + poff + i];
                x[i] = v;
            }
            s += v;
        }
        s = gsum<G>(s, red);
        const float m = s * invP;
        float q = 0.f;
        for (int i = l; i < P; i += G) { const float d = x[i] - m; q += d * d; }
        q = gsum<G>(q, red);
        const float r = 1.0f / sqrtf(q * invP + a.eps);
        mean[br] = m; rstd[br] = r;
        if (l == 0) {
            float* st = a.stats + ((long long)n * Cx + cx) * 2;
            st[0] = m; st[1] = r;
        }
    }
    const float g0 = a.gamma[0][c], b0 = a.beta[0][c];
    float g1 = 0.f, b1 = 0.f;
    if (nbr == 2) { g1 = a.gamma[1][c]; b1 = a.beta[1][c]; }
    const long long yoff0 = (long long)n * a.y_sn + (long long)c * a.y_sc;
    for (int i = l; i < P; i += G) {
        const int h = i / a.W, w = i - h * a.W;
        const float z0 = (xp[0][i] - mean[0]) * rstd[0] * g0 + b0;
        float y;
        if (a.act == ACT_GLU) {
            const float z1 = (xp[1][i] - mean[1]) * rstd[1] * g1 + b1;
            y = z0 * sigmoidf_(z1);
        } else if (a.act == ACT_SILU) {
            y = z0 * sigmoidf_(z0);
        } else {
            y = z0;
        }
        const long long yo = a.y_xs ? yoff0 + xs_off(h, w, a.xs_pw, a.xs_plane) : yoff0 + (long long)h * a.y_sh + w;
        if (a.res) y += a.res[yo];
        a.y[yo] = y;
    }
    if (a.y_xs) xs_zero_borders<G>(a.y + yoff0, a.H, a.xs_pw, a.xs_plane, l);
}

// one group per channel c, looping over n: d(gamma), d(beta) are owned by the group -> no atomics,
// deterministic accumulation order.
template <int G>
__global__ void __launch_bounds__(256) norm_bwd_kernel(const Twin<NormBwdArgs> tw)
{
    const NormBwdArgs a = tw.v[blockIdx.z];
    __shared__ float red[4];
    constexpr int GPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int c = blockIdx.x * GPB + g;
    if (c >= a.C) return;
    const int P = a.H * a.W;
    const float invP = 1.0f / (float)P;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    const int Cx = a.C * nbr;
    float gam[2] = {0.f, 0.f}, bet[2] = {0.f, 0.f};
    // (constant indices only: a run-time index into the by-value argument struct -- a.gamma[br] -- made the compiler keep ALL 216 bytes of it in
    //  scratch memory: every thread wrote its copy and read the fields back, 54 KB of private-segment traffic per workgroup -- the '6x its own
    //  bytes' the PMC passes reported for this family (r6: profiles/r06_norm_bwd_probe.log, tools/isa_scan.py `scratch` column))
    gam[0] = a.gamma[0][c]; bet[0] = a.beta[0][c];
    if (nbr == 2) { gam[1] = a.gamma[1][c]; bet[1] = a.beta[1][c]; }
    float dgam[2] = {0.f, 0.f}, dbet[2] = {0.f, 0.f};
    const int oh_w = a.W >> 1;   // conv-grid width when un-shuffling
    (void)oh_w;
    for (int n = 0; n < a.N; ++n) {
        float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
        const float* xp[2] = {nullptr, nullptr};
        for (int br = 0; br < nbr; ++br) {
            const int cx = c + br * a.C;
            const float* st = a.stats + ((long long)n * Cx + cx) * 2;
            mean[br] = st[0]; rstd[br] = st[1];
            xp[br] = a.x + (long long)n * a.x_sn + (long long)cx * a.x_sc;
        }
        const long long yoff0 = (long long)n * a.y_sn + (long long)c * a.y_sc;
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        for (int i = l; i < P; i += G) {
            const int h = i / a.W, w = i - h * a.W;
            const long long yo = yoff0 + (long long)h * a.y_sh + w;
            float dyv = a.dy[yo];
            if (a.nslab > 1) {
                for (int sl = 1; sl < a.nslab; ++sl) dyv += a.dy_slabs[(long long)(sl - 1) * a.slab_stride + yo];
                a.dy[yo] = dyv;
            }
            const float xh0 = (xp[0][i] - mean[0]) * rstd[0];
            const float z0 = xh0 * gam[0] + bet[0];
            float dz0, dz1 = 0.f, xh1 = 0.f;
            if (a.act == ACT_GLU) {
                xh1 = (xp[1][i] - mean[1]) * rstd[1];
                const float sg = sigmoidf_(xh1 * gam[1] + bet[1]);
                dz0 = dyv * sg;
                dz1 = dyv * z0 * sg * (1.0f - sg);
            } else if (a.act == ACT_SILU) {
                const float sg = sigmoidf_(z0);
                dz0 = dyv * (sg * (1.0f + z0 * (1.0f - sg)));
            } else {
                dz0 = dyv;
            }
            s1[0] += dz0; s2[0] += dz0 * xh0;
            s1[1] += dz1; s2[1] += dz1 * xh1;
        }
        for (int br = 0; br < nbr; ++br) {
            s1[br] = gsum<G>(s1[br], red);
            s2[br] = gsum<G>(s2[br], red);
            dbet[br] += s1[br];
            dgam[br] += s2[br];
        }
        for (int i = l; i < P; i += G) {
            const int h = i / a.W, w = i - h * a.W;
            const long long yo = yoff0 + (long long)h * a.y_sh + w;
            const float dyv = a.dy[yo];
            const float xh0 = (xp[0][i] - mean[0]) * rstd[0];
            const float z0 = xh0 * gam[0] + bet[0];
            float dz0, dz1 = 0.f, xh1 = 0.f;
            if (a.act == ACT_GLU) {
                xh1 = (xp[1][i] - mean[1]) * rstd[1];
                const float sg = sigmoidf_(xh1 * gam[1] + bet[1]);
                dz0 = dyv * sg;
                dz1 = dyv * z0 * sg * (1.0f - sg);
            } else if (a.act == ACT_SILU) {
                const float sg = sigmoidf_(z0);
                dz0 = dyv * (sg * (1.0f + z0 * (1.0f - sg)));
            } else {
                dz0 = dyv;
            }
            const float dx0 = gam[0] * rstd[0] * (dz0 - s1[0] * invP - xh0 * (s2[0] * invP));
            if (a.unshuffle) {
                const int co = 4 * c + 2 * (h & 1) + (w & 1);
                a.dx[(long long)n * a.dx_sn + (long long)co * a.dx_sc + (long long)(h >> 1) * a.dx_sh + (w >> 1)] = dx0;
            } else {
                const long long di = a.dx_pitch ? (long long)h * a.dx_pitch + w : i;
                a.dx[(long long)n * a.dx_sn + (long long)c * a.dx_sc + di] = dx0;
                if (nbr == 2) {
                    const float dx1 = gam[1] * rstd[1] * (dz1 - s1[1] * invP - xh1 * (s2[1] * invP));
                    a.dx[(long long)n * a.dx_sn + (long long)(c + a.C) * a.dx_sc + di] = dx1;
                }
            }
        }
        if (a.dx_pitch && !a.unshuffle)
            for (int br = 0; br < nbr; ++br) dyp_zero_borders<G>(a.dx + (long long)n * a.dx_sn + (long long)(c + br * a.C) * a.dx_sc, a.H, a.W, a.dx_pitch, l);
    }
    if (l == 0) {
        if (a.dgamma[0]) a.dgamma[0][c] += dgam[0];
        if (a.dbeta[0]) a.dbeta[0][c] += dbet[0];
        if (nbr == 2) {
            if (a.dgamma[1]) a.dgamma[1][c] += dgam[1];
            if (a.dbeta[1]) a.dbeta[1][c] += dbet[1];
        }
    }
}

// ---- register-cached variants -----------------------------------------------------------------------
// Each lane keeps its <= E plane elements in registers: ONE global read of the plane (all loads issued
// back-to-back), statistics by shuffles, one write.  At bs=1 these kernels are pure latency, so the number
// of dependent memory round trips (1 here vs 5-6 in the streaming kernels above) is what matters.
template <int G, int E>
__global__ void __launch_bounds__(256) norm_fwd_reg_kernel(const Twin<NormArgs> tw)
{
    const NormArgs a = tw.v[blockIdx.z];
    __shared__ float red[4];
    constexpr int GPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const long long plane_id = (long long)blockIdx.x * GPB + g;
    if (plane_id >= (long long)a.N * a.C) return;
    const int n = (int)(plane_id / a.C), c = (int)(plane_id - (long long)n * a.C);
    const int P = a.H * a.W;
    const float invP = 1.0f / (float)P;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    const int Cx = a.C * nbr;
    float xv[2][E];
    float rv[E];
    long long yo[E];
    // ---- issue every load first
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        if (br < nbr) {
            const long long poff = (long long)n * a.x_sn + (long long)(c + br * a.C) * a.x_sc;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = l + e * G;
                float v = 0.f;
                if (i < P) {
                    v = a.x[poff + i];
                    // (kept per element: the slab-major form of norm_bwd_reg_kernel below is the same arithmetic, but with it hipcc contracts
                    //  `x - s * invP` of the statistics differently -- a 1e-7 change per element that four Adam steps amplify to 1e-3 on single
                    //  tensors and that moved the deterministic fp64-anchor numbers of tests/test_hip_parity_fp64.py; neutral in time either way)
#pragma unroll 8
                    for (int sl = 1; sl < a.nslab; ++sl) v += a.x_slabs[(long long)(sl - 1) * a.slab_stride + poff + i];
                }
                xv[br][e] = v;
            }
        }
    }
    const long long yoff0 = (long long)n * a.y_sn + (long long)c * a.y_sc;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = l + e * G;
        const int h = i / a.W, w = i - h * a.W;
        yo[e] = a.y_xs ? yoff0 + xs_off(h, w, a.xs_pw, a.xs_plane) : yoff0 + (long long)h * a.y_sh + w;
        rv[e] = (a.res != nullptr && i < P) ? a.res[yo[e]] : 0.f;
    }
    const float g0 = a.gamma[0][c], b0 = a.beta[0][c];
    float g1 = 0.f, b1 = 0.f;
    if (nbr == 2) { g1 = a.gamma[1][c]; b1 = a.beta[1][c]; }
    // ---- statistics
    float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        if (br < nbr) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) s += xv[br][e];          // padding lanes hold 0
            s = gsum<G>(s, red);
            const float m = s * invP;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) { const float d = xv[br][e] - m; q += (l + e * G < P) ? d * d : 0.f; }
            q = gsum<G>(q, red);
            const float r = 1.0f / sqrtf(q * invP + a.eps);
            mean[br] = m; rstd[br] = r;
            if (l == 0) {
                float* st = a.stats + ((long long)n * Cx + c + br * a.C) * 2;
                st[0] = m; st[1] = r;
            }
            if (a.nslab > 1) {                                    // backward needs the reduced conv output
                const long long poff = (long long)n * a.x_sn + (long long)(c + br * a.C) * a.x_sc;
#pragma unroll
                for (int e = 0; e < E; ++e) if (l + e * G < P) a.x[poff + l + e * G] = xv[br][e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (l + e * G < P) {
            const float z0 = (xv[0][e] - mean[0]) * rstd[0] * g0 + b0;
            float y;
            if (a.act == ACT_GLU) y = z0 * sigmoidf_((xv[1][e] - mean[1]) * rstd[1] * g1 + b1);
            else if (a.act == ACT_SILU) y = z0 * sigmoidf_(z0);
            else y = z0;
            a.y[yo[e]] = y + rv[e];
        }
    }
    if (a.y_xs) xs_zero_borders<G>(a.y + yoff0, a.H, a.xs_pw, a.xs_plane, l);
}

template <int G, int E>
__global__ void __launch_bounds__(256) norm_bwd_reg_kernel(const Twin<NormBwdArgs> tw)
{
    const NormBwdArgs a = tw.v[blockIdx.z];
    __shared__ float red[4];
    constexpr int GPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const int c = blockIdx.x * GPB + g;
    if (c >= a.C) return;
    const int P = a.H * a.W;
    const float invP = 1.0f / (float)P;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    const int Cx = a.C * nbr;
    float gam[2] = {0.f, 0.f}, bet[2] = {0.f, 0.f};
    gam[0] = a.gamma[0][c]; bet[0] = a.beta[0][c];        // (constant indices: see norm_bwd_kernel)
    if (nbr == 2) { gam[1] = a.gamma[1][c]; bet[1] = a.beta[1][c]; }
    float dgam[2] = {0.f, 0.f}, dbet[2] = {0.f, 0.f};
    // blockIdx.y = a chunk of samples (large batches: the per-sample work is independent, only dgamma / dbeta are sums over samples;
    // with more than one chunk they are added with atomics -- the launcher uses one chunk in deterministic mode)
    const int nchunk = (int)gridDim.y;
    const int n_per = (a.N + nchunk - 1) / nchunk;
    const int n_begin = (int)blockIdx.y * n_per;
    const int n_end = (n_begin + n_per < a.N) ? n_begin + n_per : a.N;
    for (int n = n_begin; n < n_end; ++n) {
        float xh[2][E], dz[2][E], dyv[E];
        int hh[E], ww[E];
        const long long yoff0 = (long long)n * a.y_sn + (long long)c * a.y_sc;
        float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            if (br < nbr) {
                const float* st = a.stats + ((long long)n * Cx + c + br * a.C) * 2;
                mean[br] = st[0]; rstd[br] = st[1];
                const float* xp = a.x + (long long)n * a.x_sn + (long long)(c + br * a.C) * a.x_sc;
#pragma unroll
                for (int e = 0; e < E; ++e) { const int i = l + e * G; xh[br][e] = (i < P) ? xp[i] : 0.f; }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = l + e * G;
            const int h = i / a.W, w = i - h * a.W;
            hh[e] = h; ww[e] = w;
            dyv[e] = (i < P) ? a.dy[yoff0 + (long long)h * a.y_sh + w] : 0.f;
        }
        if (a.nslab > 1) {                     // slab-major, a round's loads independent (see norm_fwd_reg_kernel); same summation order
            for (int sl = 1; sl < a.nslab; ++sl) {
                const float* sp = a.dy_slabs + (long long)(sl - 1) * a.slab_stride + yoff0;
                float t[E];
#pragma unroll
                for (int e = 0; e < E; ++e) t[e] = (l + e * G < P) ? sp[(long long)hh[e] * a.y_sh + ww[e]] : 0.f;
#pragma unroll
                for (int e = 0; e < E; ++e) dyv[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (l + e * G < P) a.dy[yoff0 + (long long)hh[e] * a.y_sh + ww[e]] = dyv[e];      // the residual path re-reads the reduced gradient
        }
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = (l + e * G) < P;
            const float x0 = (xh[0][e] - mean[0]) * rstd[0];
            const float z0 = x0 * gam[0] + bet[0];
            float d0, d1 = 0.f, x1 = 0.f;
            if (a.act == ACT_GLU) {
                x1 = (xh[1][e] - mean[1]) * rstd[1];
                const float sg = sigmoidf_(x1 * gam[1] + bet[1]);
                d0 = dyv[e] * sg;
                d1 = dyv[e] * z0 * sg * (1.0f - sg);
            } else if (a.act == ACT_SILU) {
                const float sg = sigmoidf_(z0);
                d0 = dyv[e] * (sg * (1.0f + z0 * (1.0f - sg)));
            } else {
                d0 = dyv[e];
            }
            if (!ok) { d0 = 0.f; d1 = 0.f; }
            xh[0][e] = ok ? x0 : 0.f; xh[1][e] = ok ? x1 : 0.f;
            dz[0][e] = d0; dz[1][e] = d1;
            s1[0] += d0; s2[0] += d0 * xh[0][e];
            s1[1] += d1; s2[1] += d1 * xh[1][e];
        }
        for (int br = 0; br < nbr; ++br) {
            s1[br] = gsum<G>(s1[br], red);
            s2[br] = gsum<G>(s2[br], red);
            dbet[br] += s1[br];
            dgam[br] += s2[br];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = l + e * G;
            if (i < P) {
                const float dx0 = gam[0] * rstd[0] * (dz[0][e] - s1[0] * invP - xh[0][e] * (s2[0] * invP));
                if (a.unshuffle) {
                    const int h = hh[e], w = ww[e];
                    const int co = 4 * c + 2 * (h & 1) + (w & 1);
                    a.dx[(long long)n * a.dx_sn + (long long)co * a.dx_sc + (long long)(h >> 1) * a.dx_sh + (w >> 1)] = dx0;
                } else {
                    const long long di = a.dx_pitch ? (long long)hh[e] * a.dx_pitch + ww[e] : i;
                    a.dx[(long long)n * a.dx_sn + (long long)c * a.dx_sc + di] = dx0;
                    if (nbr == 2) {
                        const float dx1 = gam[1] * rstd[1] * (dz[1][e] - s1[1] * invP - xh[1][e] * (s2[1] * invP));
                        a.dx[(long long)n * a.dx_sn + (long long)(c + a.C) * a.dx_sc + di] = dx1;
                    }
                }
            }
        }
        if (a.dx_pitch && !a.unshuffle)
            for (int br = 0; br < nbr; ++br) dyp_zero_borders<G>(a.dx + (long long)n * a.dx_sn + (long long)(c + br * a.C) * a.dx_sc, a.H, a.W, a.dx_pitch, l);
    }
    if (l == 0 && n_begin < n_end) {
        float* const dgp[2] = {a.dgamma[0], a.dgamma[1]};
        float* const dbp[2] = {a.dbeta[0], a.dbeta[1]};
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            if (br < nbr) {
                if (nchunk > 1) {
                    if (dgp[br]) unsafeAtomicAdd(&dgp[br][c], dgam[br]);
                    if (dbp[br]) unsafeAtomicAdd(&dbp[br][c], dbet[br]);
                } else {
                    if (dgp[br]) dgp[br][c] += dgam[br];
                    if (dbp[br]) dbp[br][c] += dbet[br];
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) act_fwd_kernel(const Twin<ActArgs> tw)
{
    const ActArgs a = tw.v[blockIdx.z];
    const long long total = (long long)a.N * a.C * a.P;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % a.P);
        const long long nc = idx / a.P;
        const int c = (int)(nc % a.C);
        const long long n = nc / a.C;
        float v[2] = {0.f, 0.f};
        for (int br = 0; br < nbr; ++br) {
            const long long xo = ((n * nbr * a.C) + c + br * a.C) * a.P + i;
            float t = a.x[xo];
            if (a.nslab > 1) {
#pragma unroll 8
                for (int sl = 1; sl < a.nslab; ++sl) t += a.x_slabs[(long long)(sl - 1) * a.slab_stride + xo];
                a.x[xo] = t;
            }
            v[br] = t;
        }
        float y;
        if (a.act == ACT_GLU) y = v[0] * sigmoidf_(v[1]);
        else if (a.act == ACT_SILU) y = v[0] * sigmoidf_(v[0]);
        else if (a.act == ACT_SIGMOID) y = sigmoidf_(v[0]);
        else y = v[0];
        if (a.y) a.y[idx] = y;
    }
}

__global__ void __launch_bounds__(256) act_bwd_kernel(const Twin<ActBwdArgs> tw)
{
    const ActBwdArgs a = tw.v[blockIdx.z];
    const long long total = (long long)a.N * a.C * a.P;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % a.P);
        const long long nc = idx / a.P;
        const int c = (int)(nc % a.C);
        const long long n = nc / a.C;
        float dyv = a.dy[idx];
        if (a.nslab > 1) {
#pragma unroll 8
            for (int sl = 1; sl < a.nslab; ++sl) dyv += a.dy_slabs[(long long)(sl - 1) * a.slab_stride + idx];
            a.dy[idx] = dyv;
        }
        const long long xo0 = ((n * nbr * a.C) + c) * a.P + i;
        const float x0 = a.x[xo0];
        if (a.act == ACT_GLU) {
            const long long xo1 = xo0 + (long long)a.C * a.P;
            const float sg = sigmoidf_(a.x[xo1]);
            a.dx[xo0] = dyv * sg;
            a.dx[xo1] = dyv * x0 * sg * (1.0f - sg);
        } else if (a.act == ACT_SILU) {
            const float sg = sigmoidf_(x0);
            a.dx[xo0] = dyv * (sg * (1.0f + x0 * (1.0f - sg)));
        } else if (a.act == ACT_SIGMOID) {
            const float sg = sigmoidf_(x0);
            a.dx[xo0] = dyv * sg * (1.0f - sg);
        } else {
            a.dx[xo0] = dyv;
        }
    }
}

// 16-byte form of the two kernels above for planes of 4k elements (every layer of the network): one index division per FOUR elements in
// 32-bit arithmetic -- the scalar form spent its time in 64-bit divisions (conv1's GLU: 9.3 us for 7.9 MB)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// t + slab 1 + slab 2 + ... (that order), four slabs' loads in flight at a time: `t = add4(t, ld4(slab sl))` in a loop made every load wait for
// the previous one (one register for the loaded value, vmcnt(0) behind each load -- r6, tools/isa_scan.py); past the last slab the load
// repeats the last one and its add is skipped
__device__ __forceinline__ float4 slab_sum4(float4 t, const float* slabs, long long stride, long long off, int nslab)
{
    for (int sl = 1; sl < nslab; sl += 4) {
        float4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int s_ = (sl + k < nslab) ? sl + k : nslab - 1; u[k] = ld4(slabs + (long long)(s_ - 1) * stride + off); }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (sl + k < nslab) t = add4(t, u[k]);
    }
    return t;
}

__global__ void __launch_bounds__(256) act_fwd_vec_kernel(const Twin<ActArgs> tw)
{
    const ActArgs a = tw.v[blockIdx.z];
    const unsigned p4 = (unsigned)a.P >> 2;
    const unsigned total4 = (unsigned)a.N * (unsigned)a.C * p4;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total4; q += gridDim.x * 256u) {
        const unsigned nc = q / p4, i4 = q - nc * p4;
        const unsigned n = nc / (unsigned)a.C, c = nc - n * (unsigned)a.C;
        float4 v[2];
        v[1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int br = 0; br < 2; ++br) {                     // (fully unrolled: v[] must stay in registers)
            if (br < nbr) {
                const long long xo = ((long long)(n * nbr * a.C + c + br * a.C)) * a.P + 4 * i4;
                float4 t = ld4(a.x + xo);
                if (a.nslab > 1) {
                    t = slab_sum4(t, a.x_slabs, a.slab_stride, xo, a.nslab);
                    st4(a.x + xo, t);
                }
                v[br] = t;
            }
        }
        float4 y;
        if (a.act == ACT_GLU) y = make_float4(v[0].x * sigmoidf_(v[1].x), v[0].y * sigmoidf_(v[1].y), v[0].z * sigmoidf_(v[1].z), v[0].w * sigmoidf_(v[1].w));
        else if (a.act == ACT_SILU) y = make_float4(v[0].x * sigmoidf_(v[0].x), v[0].y * sigmoidf_(v[0].y), v[0].z * sigmoidf_(v[0].z), v[0].w * sigmoidf_(v[0].w));
        else if (a.act == ACT_SIGMOID) y = make_float4(sigmoidf_(v[0].x), sigmoidf_(v[0].y), sigmoidf_(v[0].z), sigmoidf_(v[0].w));
        else y = v[0];
        if (a.y) st4(a.y + 4LL * q, y);
    }
}

__global__ void __launch_bounds__(256) act_bwd_vec_kernel(const Twin<ActBwdArgs> tw)
{
    const ActBwdArgs a = tw.v[blockIdx.z];
    const unsigned p4 = (unsigned)a.P >> 2;
    const unsigned total4 = (unsigned)a.N * (unsigned)a.C * p4;
    const int nbr = (a.act == ACT_GLU) ? 2 : 1;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total4; q += gridDim.x * 256u) {
        const unsigned nc = q / p4, i4 = q - nc * p4;
        const unsigned n = nc / (unsigned)a.C, c = nc - n * (unsigned)a.C;
        float4 d = ld4(a.dy + 4LL * q);
        if (a.nslab > 1) {
            d = slab_sum4(d, a.dy_slabs, a.slab_stride, 4LL * q, a.nslab);
            st4(a.dy + 4LL * q, d);
        }
        const long long xo0 = ((long long)(n * nbr * a.C + c)) * a.P + 4 * i4;
        const float4 x0 = ld4(a.x + xo0);
        const float dv[4] = {d.x, d.y, d.z, d.w}, xv[4] = {x0.x, x0.y, x0.z, x0.w};
        float o0[4], o1[4];
        if (a.act == ACT_GLU) {
            const long long xo1 = xo0 + (long long)a.C * a.P;
            const float4 x1 = ld4(a.x + xo1);
            const float gv[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float sg = sigmoidf_(gv[e]); o0[e] = dv[e] * sg; o1[e] = dv[e] * xv[e] * sg * (1.0f - sg); }
            st4(a.dx + xo0, make_float4(o0[0], o0[1], o0[2], o0[3]));
            st4(a.dx + xo1, make_float4(o1[0], o1[1], o1[2], o1[3]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a.act == ACT_SILU) { const float sg = sigmoidf_(xv[e]); o0[e] = dv[e] * (sg * (1.0f + xv[e] * (1.0f - sg))); }
                else if (a.act == ACT_SIGMOID) { const float sg = sigmoidf_(xv[e]); o0[e] = dv[e] * sg * (1.0f - sg); }
                else o0[e] = dv[e];
            }
            st4(a.dx + xo0, make_float4(o0[0], o0[1], o0[2], o0[3]));
        }
    }
}

// Output transform of a Winograd convolution + instance norm + activation in one pass: the products M[xi][co][tile] are turned into the
// S x S outputs of each tile in registers (+ bias), the plane statistics are taken over them, and both the conv output (the backward pass
// reads it) and the normalised / activated plane are stored -- the separate output-transform launch and the norm's re-read of the conv
// output go away (4 launches per generator forward: downSample1/2, upSample1/2, model.py:245-246, 274-275).
//   PTS = 16: F(2x2,3x3) (the stride-2 5x5 layers in phase form), plane (n, c) = conv channel c (value) and C + c (gate, GLU)
//   PTS = 36: F(2x2,5x5) with PixelShuffle(2): plane (n, c) = conv channels 4c .. 4c+3 interleaved; items = (sub-channel, tile)
//   PTS = 43: F(4x4,3x3) (36 points, 4x4 outputs), planes as PTS = 16          PTS = 64: F(4x4,5x5) (4x4 outputs), planes as PTS = 36
// G threads share one plane, every thread owns TT items.
template <int PTS> struct WinoOutT;
template <> struct WinoOutT<16> { static constexpr int R = 4, S = 2; static constexpr bool SHUF = false;
    static __device__ __forceinline__ void at(const float* m, float* o) { o[0] = m[0] + m[1] + m[2]; o[1] = m[1] - m[2] - m[3]; } };
template <> struct WinoOutT<36> { static constexpr int R = 6, S = 2; static constexpr bool SHUF = true;
    static __device__ __forceinline__ void at(const float* m, float* o) { o[0] = m[0] + m[1] + m[2] + m[3] + m[4]; o[1] = m[1] - m[2] + 2.f * (m[3] - m[4]) + m[5]; } };
template <> struct WinoOutT<43> { static constexpr int R = 6, S = 4; static constexpr bool SHUF = false;
    static __device__ __forceinline__ void at(const float* m, float* o)
    {
        const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        o[0] = m[0] + s12 + s34; o[1] = d12 + 2.f * d34; o[2] = s12 + 4.f * s34; o[3] = d12 + 8.f * d34 + m[5];
    } };
template <> struct WinoOutT<64> { static constexpr int R = 8, S = 4; static constexpr bool SHUF = true;
    static __device__ __forceinline__ void at(const float* m, float* o)          // A^T of wino4.h: points {0, 1, -1, 2, -2, 1/2, -1/2, inf}
    {
        const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
        o[0] = m[0] + s12 + s34 + s56;
        o[1] = d12 + 2.f * d34 + 0.5f * d56;
        o[2] = s12 + 4.f * s34 + 0.25f * s56;
        o[3] = d12 + 8.f * d34 + 0.125f * d56 + m[7];
    } };

struct NormFwdWinoKArgs { NormArgs a; WinoOutArgs w; };
template <int G, int TT, int PTS>
__global__ void __launch_bounds__(256) norm_fwd_wino_kernel(const Twin<NormFwdWinoKArgs> tw)
{
    const NormFwdWinoKArgs ka_ = tw.v[blockIdx.z];
    const NormArgs& a = ka_.a;
    const WinoOutArgs& w = ka_.w;
    __shared__ float red[4];
    using WT = WinoOutT<PTS>;
    constexpr bool SHUF = WT::SHUF;
    constexpr int R = WT::R, S = WT::S, E = S * S;
    constexpr int GPB = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    const long long plane_id = (long long)blockIdx.x * GPB + g;
    if (plane_id >= (long long)a.N * a.C) return;
    const int n = (int)(plane_id / a.C), c = (int)(plane_id - (long long)n * a.C);
    const int tiles = w.TH * w.TW;
    const int items = SHUF ? 4 * tiles : tiles;
    const int nbr = (!SHUF && a.act == ACT_GLU) ? 2 : 1;
    const int Cx = a.C * nbr;
    const long long xs = (long long)w.Cout * w.NTp;
    const float invP = 1.0f / (float)(a.H * a.W);
    float xv[2][TT][E];
    int pos[TT];                                     // plane offset h * W + w of the item's first output (-1: no item)
#pragma unroll
    for (int k = 0; k < TT; ++k) {
        const int item = l + k * G;
        const int sub = SHUF ? item / tiles : 0;
        const int t = item - sub * tiles;
        const int ty = t / w.TW, tx = t - ty * w.TW;
        pos[k] = item < items ? (SHUF ? (2 * S * ty + (sub >> 1)) * a.W + 2 * S * tx + (sub & 1) : S * ty * a.W + S * tx) : -1;
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            if (br < nbr) {
                const int co = SHUF ? 4 * c + sub : c + br * a.C;
                float o[E];
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = 0.f;
                if (item < items) {
                    const float* src = w.m + (long long)co * w.NTp + (long long)n * tiles + t;
                    float u[S][R];
#pragma unroll
                    for (int b = 0; b < R; ++b) {
                        float m[R], q[S];
#pragma unroll
                        for (int aa = 0; aa < R; ++aa) m[aa] = src[(long long)(aa * R + b) * xs];
                        WT::at(m, q);
#pragma unroll
                        for (int i = 0; i < S; ++i) u[i][b] = q[i];
                    }
                    const float bias = w.bias ? w.bias[co] : 0.f;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        float q[S];
                        WT::at(u[i], q);
#pragma unroll
                        for (int j = 0; j < S; ++j) o[i * S + j] = q[j] + bias;
                    }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) xv[br][k][e] = o[e];
            }
        }
    }
    const float g0 = a.gamma[0][c], b0 = a.beta[0][c];
    float g1 = 0.f, b1 = 0.f;
    if (nbr == 2) { g1 = a.gamma[1][c]; b1 = a.beta[1][c]; }
    // outputs of one item: rows dh apart, columns dw apart in the plane
    const int dh = (SHUF ? 2 : 1) * a.W, dw = SHUF ? 2 : 1;
    float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        if (br < nbr) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < TT; ++k) {
                float sk = 0.f;
#pragma unroll
                for (int e = 0; e < E; e += 4) sk += (xv[br][k][e] + xv[br][k][e + 1]) + (xv[br][k][e + 2] + xv[br][k][e + 3]);
                s += sk;
            }
            s = gsum<G>(s, red);
            const float m = s * invP;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < TT; ++k) {
                if (pos[k] >= 0) {
#pragma unroll
                    for (int e = 0; e < E; ++e) { const float d = xv[br][k][e] - m; q += d * d; }
                }
            }
            q = gsum<G>(q, red);
            const float r = 1.0f / sqrtf(q * invP + a.eps);
            mean[br] = m; rstd[br] = r;
            if (l == 0) {
                float* st = a.stats + ((long long)n * Cx + c + br * a.C) * 2;
                st[0] = m; st[1] = r;
            }
            float* xp = a.x + (long long)n * a.x_sn + (long long)(c + br * a.C) * a.x_sc;
#pragma unroll
            for (int k = 0; k < TT; ++k) {
                if (pos[k] >= 0) {
#pragma unroll
                    for (int e = 0; e < E; ++e) xp[pos[k] + (e / S) * dh + (e % S) * dw] = xv[br][k][e];
                }
            }
        }
    }
    float* yp = a.y + (long long)n * a.y_sn + (long long)c * a.y_sc;
#pragma unroll
    for (int k = 0; k < TT; ++k) {
        if (pos[k] >= 0) {
            const int h0 = pos[k] / a.W, w0 = pos[k] - h0 * a.W;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float z0 = (xv[0][k][e] - mean[0]) * rstd[0] * g0 + b0;
                float y;
                if (a.act == ACT_GLU) y = z0 * sigmoidf_((xv[1][k][e] - mean[1]) * rstd[1] * g1 + b1);
                else if (a.act == ACT_SILU) y = z0 * sigmoidf_(z0);
                else y = z0;
                yp[(long long)(h0 + (e / S) * (SHUF ? 2 : 1)) * a.y_sh + w0 + (e % S) * dw] = y;
            }
        }
    }
}

static int pick_group(int P) { return P <= 32 ? 16 : (P <= 640 ? 64 : 256); }

}  // namespace

// items per plane the fused output-transform + norm kernel would see, or 0 when the pair does not fit it
static int wino_norm_items(const NormArgs& a, const WinoOutArgs& w, int pts)
{
    if (a.nslab != 1 || a.res != nullptr || w.accumulate || w.N != a.N || (w.OH & 1) || (w.OW & 1) || w.NT != w.N * w.TH * w.TW) return 0;
    if ((pts == 43 || pts == 64) && ((w.OH & 3) || (w.OW & 3))) return 0;
    if (pts == 16 || pts == 43) {
        if (w.shuffle || a.H != w.OH || a.W != w.OW || w.Cout != a.C * (a.act == ACT_GLU ? 2 : 1) || a.x_sc != (long long)a.H * a.W) return 0;
        return w.TH * w.TW;
    }
    if (pts == 36 || pts == 64) {
        if (!w.shuffle || a.act == ACT_GLU || a.H != 2 * w.OH || a.W != 2 * w.OW || w.Cout != 4 * a.C || a.x_sc != (long long)a.H * a.W) return 0;
        return 4 * w.TH * w.TW;
    }
    return 0;
}
// (registers: an item holds 4 outputs at 2x2 tiles, 16 at 4x4 -- times two branches with GLU)
static int wino_norm_max_items(int pts) { return (pts == 43 || pts == 64) ? 320 : 1280; }

bool mcvc_norm_fwd_wino_applies(const NormArgs& a, const WinoOutArgs& w, int pts)
{
    const int items = wino_norm_items(a, w, pts);
    return items > 0 && items <= wino_norm_max_items(pts);
}

int mcvc_norm_fwd_wino_launch(const NormArgs& a, const WinoOutArgs& w, int pts, hipStream_t s)
{
    const int items = wino_norm_items(a, w, pts);
    if (items <= 0 || items > wino_norm_max_items(pts)) return MCVC_ERR_INVALID;
    const long long planes = (long long)a.N * a.C;
    const int nbr = ((pts == 16 || pts == 43) && a.act == ACT_GLU) ? 2 : 1;
    const double el = (double)planes * a.H * a.W * nbr;
    const double m_per_out = pts == 16 ? 4.0 : (pts == 36 ? 9.0 : (pts == 43 ? 2.25 : 4.0));      // points per output
    TraceScope ts(K_NORM_FWD, s, 0.0, 4.0 * (el * m_per_out + el + (double)planes * a.H * a.W));
#define MCVC_FWD_WINO(GG, TT) { \
        if (pts == 16) mcvc_launch((norm_fwd_wino_kernel<GG, TT, 16>), dim3((unsigned)cdiv_ll(planes, 256 / GG)), dim3(256), 0, s, NormFwdWinoKArgs{a, w}); \
        else if (pts == 36) mcvc_launch((norm_fwd_wino_kernel<GG, TT, 36>), dim3((unsigned)cdiv_ll(planes, 256 / GG)), dim3(256), 0, s, NormFwdWinoKArgs{a, w}); \
        else if (pts == 43) mcvc_launch((norm_fwd_wino_kernel<GG, TT, 43>), dim3((unsigned)cdiv_ll(planes, 256 / GG)), dim3(256), 0, s, NormFwdWinoKArgs{a, w}); \
        else mcvc_launch((norm_fwd_wino_kernel<GG, TT, 64>), dim3((unsigned)cdiv_ll(planes, 256 / GG)), dim3(256), 0, s, NormFwdWinoKArgs{a, w}); \
        return (int)hipGetLastError(); }
    if (pts == 43 || pts == 64) {              // 4x4 outputs per item: a quarter of the items of the 2x2 schemes on the same plane
        if (items <= 64) MCVC_FWD_WINO(64, 1)
        if (items <= 128) MCVC_FWD_WINO(64, 2)
        if (items <= 256 && planes < 2048) MCVC_FWD_WINO(256, 1)
        // (16 outputs per item: five items per thread are 220 registers -- two waves per SIMD -- and measured 210 us against 160 for the
        // separate transform + norm at 32 samples; two items per thread keep four waves resident)
        if (items <= 512) MCVC_FWD_WINO(256, 2)
        MCVC_FWD_WINO(64, 5)
    }
    if (items <= 128) MCVC_FWD_WINO(64, 2)
    if (items <= 512 && planes < 2048) MCVC_FWD_WINO(256, 2)
    if (items <= 320) MCVC_FWD_WINO(64, 5)
    MCVC_FWD_WINO(256, 5)
#undef MCVC_FWD_WINO
}

int mcvc_norm_fwd_launch(const NormArgs& a, hipStream_t s)
{
    const int P = a.H * a.W;
    const long long planes = (long long)a.N * a.C;
    const int G = pick_group(P);
    const unsigned blocks = (unsigned)cdiv_ll(planes, 256 / G);
    const double el = (double)planes * P * (a.act == ACT_GLU ? 2 : 1);
    TraceScope ts(K_NORM_FWD, s, 0.0, 4.0 * (el * (a.nslab + 1) + (double)planes * P));
#define MCVC_FWD_REG(GG, EE) { mcvc_launch((norm_fwd_reg_kernel<GG, EE>), dim3((unsigned)cdiv_ll(planes, 256 / GG)), dim3(256), 0, s, a); return (int)hipGetLastError(); }
    if (P <= 16) MCVC_FWD_REG(16, 1)
    if (P <= 64) MCVC_FWD_REG(16, 4)
    if (P <= 128) MCVC_FWD_REG(64, 2)
    if (P <= 320) MCVC_FWD_REG(64, 5)
    if (P <= 1280) MCVC_FWD_REG(256, 5)
    if (P <= 5120) MCVC_FWD_REG(256, 20)
#undef MCVC_FWD_REG
    if (G == 16) mcvc_launch(norm_fwd_kernel<16>, dim3(blocks), dim3(256), 0, s, a);
    else if (G == 64) mcvc_launch(norm_fwd_kernel<64>, dim3(blocks), dim3(256), 0, s, a);
    else mcvc_launch(norm_fwd_kernel<256>, dim3(blocks), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_norm_bwd_launch(const NormBwdArgs& a, hipStream_t s)
{
    const int P = a.H * a.W;
    const int G = pick_group(P);
    const unsigned blocks = (unsigned)cdiv_i(a.C, 256 / G);
    const double el = (double)a.N * a.C * P;
    TraceScope ts(K_NORM_BWD, s, 0.0, 4.0 * el * ((a.act == ACT_GLU ? 4 : 2) + a.nslab));
    // sample chunks: enough workgroups for the chip when the channel blocks alone are few (one chunk = the deterministic order)
    auto chunks = [&](int cblocks) {
        int nc = 1;
        if (!mcvc_deterministic())
            while (2 * nc <= a.N && nc < 32 && cblocks * nc < 768) nc *= 2;
        return nc;
    };
#define MCVC_BWD_REG(GG, EE) { const int cb = cdiv_i(a.C, 256 / GG); mcvc_launch((norm_bwd_reg_kernel<GG, EE>), dim3((unsigned)cb, (unsigned)chunks(cb)), dim3(256), 0, s, a); return (int)hipGetLastError(); }
    if (P <= 16) MCVC_BWD_REG(16, 1)
    if (P <= 64) MCVC_BWD_REG(16, 4)
    if (P <= 128) MCVC_BWD_REG(64, 2)
    if (P <= 320) MCVC_BWD_REG(64, 5)
    if (P <= 1280) MCVC_BWD_REG(256, 5)
    if (P <= 5120) MCVC_BWD_REG(256, 20)
#undef MCVC_BWD_REG
    if (G == 16) mcvc_launch(norm_bwd_kernel<16>, dim3(blocks), dim3(256), 0, s, a);
    else if (G == 64) mcvc_launch(norm_bwd_kernel<64>, dim3(blocks), dim3(256), 0, s, a);
    else mcvc_launch(norm_bwd_kernel<256>, dim3(blocks), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

static unsigned ew_blocks(long long total)
{
    long long b = cdiv_ll(total, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int mcvc_act_fwd_launch(const ActArgs& a, hipStream_t s)
{
    TraceScope ts(K_ACT_FWD, s, 0.0, 4.0 * (double)a.N * a.C * a.P * ((a.act == ACT_GLU ? 2 : 1) * a.nslab + 1));
    const long long total = (long long)a.N * a.C * a.P;
    const bool al = ((((uintptr_t)a.x | (uintptr_t)a.x_slabs | (uintptr_t)a.y) & 15) == 0) && (a.slab_stride & 3) == 0;
    if ((a.P & 3) == 0 && al && total < (1LL << 33)) mcvc_launch(act_fwd_vec_kernel, dim3(ew_blocks(total >> 2)), dim3(256), 0, s, a);
    else mcvc_launch(act_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_act_bwd_launch(const ActBwdArgs& a, hipStream_t s)
{
    TraceScope ts(K_ACT_BWD, s, 0.0, 4.0 * (double)a.N * a.C * a.P * ((a.act == ACT_GLU ? 4 : 2) + a.nslab));
    const long long total = (long long)a.N * a.C * a.P;
    const bool al = ((((uintptr_t)a.x | (uintptr_t)a.dy | (uintptr_t)a.dy_slabs | (uintptr_t)a.dx) & 15) == 0) && (a.slab_stride & 3) == 0;
    if ((a.P & 3) == 0 && al && total < (1LL << 33)) mcvc_launch(act_bwd_vec_kernel, dim3(ew_blocks(total >> 2)), dim3(256), 0, s, a);
    else mcvc_launch(act_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
