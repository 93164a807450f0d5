// Small HBM-bound kernels of the training step: FIF input conditioning, loss reductions with their
// gradients, bias gradients, flat Adam, axpy.
//
// Reference call sites: model.py:241 (stack(x*mask, mask)); train.py:219-237 and :276-294 (L1 and
// LSGAN means); train.py:119-122, 242, 299 (torch.optim.Adam, betas (0.5, 0.999), eps 1e-8).
#include "mcvc_common.h"
#include "misc.h"
#include "trace.h"
#include "launch.h"

namespace {

__device__ __forceinline__ float block_sum_1024(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// xin[n][0] = x*mask ; xin[n][1] = mask          (model.py:241)
struct PrepInputKArgs { const float* x; const float* mask; float* xin; int N; int P; };
__global__ void prep_input_kernel(const Twin<PrepInputKArgs> tw)
{
    const PrepInputKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ x = ka_.x;
    const float* __restrict__ mask = ka_.mask;
    float* __restrict__ xin = ka_.xin;
    int N = ka_.N;
    int P = ka_.P;
    const long long total = (long long)N * P;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / P;
        const int i = (int)(idx - n * P);
        const float m = mask ? mask[idx] : 1.0f;
        xin[(n * 2) * P + i] = x[idx] * m;
        xin[(n * 2 + 1) * P + i] = m;
    }
}

// dx[n][i] (+)= mask * sum_slabs dxin[n][0][i]
struct MaskGradKArgs { const float* dxin; const float* dxin_slabs; long long slab_stride; int nslab; const float* mask; float* dx; int N; int P; int C; int accumulate; };
__global__ void mask_grad_kernel(const Twin<MaskGradKArgs> tw)
{
    const MaskGradKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ dxin = ka_.dxin;
    const float* __restrict__ dxin_slabs = ka_.dxin_slabs;
    long long slab_stride = ka_.slab_stride;
    int nslab = ka_.nslab;
    const float* __restrict__ mask = ka_.mask;
    float* __restrict__ dx = ka_.dx;
    int N = ka_.N;
    int P = ka_.P;
    int C = ka_.C;
    int accumulate = ka_.accumulate;
    const long long total = (long long)N * P;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / P;
        const int i = (int)(idx - n * P);
        const long long so = (n * C) * P + i;
        float v = dxin[so];
        for (int sl = 1; sl < nslab; sl += 4) {      // four slabs' loads in flight (a `v += slab[sl]` loop waits for each load: norm_kernels.hip slab_sum4); same order
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int s_ = (sl + k < nslab) ? sl + k : nslab - 1; u[k] = dxin_slabs[(long long)(s_ - 1) * slab_stride + so]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (sl + k < nslab) v += u[k];
        }
        if (mask) v *= mask[idx];
        dx[idx] = accumulate ? dx[idx] + v : v;
    }
}

// db[c] += sum_{n,i} dy[n][c][i]      (one block per channel)
// db[c] += sum over (n, pixel) of dy: one workgroup of 1024 threads per channel, 16-byte loads, four independent partial sums per thread
// (a 256-thread scalar loop took 200-500 us per launch at 32-64 samples: the Cout = 1 layers reduce 330 k elements in ONE workgroup).
// Fixed summation order: deterministic.
struct BiasGradKArgs { const float* dy; long long sn; long long sc; int N; int P; float* db; };
__global__ void __launch_bounds__(1024) bias_grad_kernel(const Twin<BiasGradKArgs> tw)
{
    const BiasGradKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ dy = ka_.dy;
    long long sn = ka_.sn;
    long long sc = ka_.sc;
    int N = ka_.N;
    int P = ka_.P;
    float* __restrict__ db = ka_.db;
    __shared__ float red[16];
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const bool vec = ((P & 3) == 0) && ((sn & 3) == 0) && ((sc & 3) == 0) && ((reinterpret_cast<unsigned long long>(dy) & 15ull) == 0);
    if (vec) {
        const int P4 = P >> 2;
        for (int n = 0; n < N; ++n) {
            const float4* p = reinterpret_cast<const float4*>(dy + (long long)n * sn + (long long)c * sc);
            int i = threadIdx.x;
            for (; i + 3 * 1024 < P4; i += 4 * 1024) {
                const float4 a = p[i], b = p[i + 1024], e = p[i + 2048], f = p[i + 3072];
                s0 += (a.x + a.y) + (a.z + a.w); s1 += (b.x + b.y) + (b.z + b.w);
                s2 += (e.x + e.y) + (e.z + e.w); s3 += (f.x + f.y) + (f.z + f.w);
            }
            for (; i < P4; i += 1024) { const float4 a = p[i]; s0 += (a.x + a.y) + (a.z + a.w); }
        }
    } else {
        for (int n = 0; n < N; ++n) {
            const float* p = dy + (long long)n * sn + (long long)c * sc;
            for (int i = threadIdx.x; i < P; i += 1024) s0 += p[i];
        }
    }
    const float s = block_sum_1024((s0 + s1) + (s2 + s3), red);
    if (threadIdx.x == 0) db[c] += s;
}

// L1: loss[slot] += weight * mean|a - b| ;  grad_a (=/+=) weight * sign(a - b) / n     (single block, deterministic)
struct L1LossKArgs { const float* a; const float* b; long long n; float weight; float* loss_slot; float* term_slot; float* grad_a; int accumulate; };
__global__ void __launch_bounds__(1024) l1_loss_kernel(const Twin<L1LossKArgs> tw)
{
    const L1LossKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ a = ka_.a;
    const float* __restrict__ b = ka_.b;
    long long n = ka_.n;
    float weight = ka_.weight;
    float* __restrict__ loss_slot = ka_.loss_slot;
    float* __restrict__ term_slot = ka_.term_slot;
    float* __restrict__ grad_a = ka_.grad_a;
    int accumulate = ka_.accumulate;
    __shared__ float red[16];
    float s = 0.f;
    const float gs = weight / (float)n;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float d = a[i] - b[i];
        s += fabsf(d);
        if (grad_a) {
            const float g = (d > 0.f) ? gs : ((d < 0.f) ? -gs : 0.f);
            grad_a[i] = accumulate ? grad_a[i] + g : g;
        }
    }
    s = block_sum_1024(s, red);
    if (threadIdx.x == 0) {
        const float m = s / (float)n;
        if (term_slot) *term_slot += m;
        if (loss_slot) *loss_slot += weight * m;
    }
}

// LSGAN on the discriminator's sigmoid output d: loss += weight*mean((target-d)^2);
// grad wrt the PRE-sigmoid logit: weight * 2 (d - target)/n * d (1-d)
struct LsganLossKArgs { const float* d; long long n; float target; float weight; float* loss_slot; float* term_slot; float* grad_logit; };
__global__ void __launch_bounds__(1024) lsgan_loss_kernel(const Twin<LsganLossKArgs> tw)
{
    const LsganLossKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ d = ka_.d;
    long long n = ka_.n;
    float target = ka_.target;
    float weight = ka_.weight;
    float* __restrict__ loss_slot = ka_.loss_slot;
    float* __restrict__ term_slot = ka_.term_slot;
    float* __restrict__ grad_logit = ka_.grad_logit;
    __shared__ float red[16];
    float s = 0.f;
    const float gs = 2.0f * weight / (float)n;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float v = d[i];
        const float e = v - target;
        s += e * e;
        if (grad_logit) grad_logit[i] = gs * e * v * (1.0f - v);
    }
    s = block_sum_1024(s, red);
    if (threadIdx.x == 0) {
        const float m = s / (float)n;
        if (term_slot) *term_slot += m;
        if (loss_slot) *loss_slot += weight * m;
    }
}

// The loss terms of one phase are produced on different streams, each into a PRIVATE pair (weighted value, plain mean); this single
// thread adds them to the public slots in a FIXED order (the reference's: train.py:233-237, 276-294), so the sums are reproducible
// whatever the streams' relative timing was.
struct LossCombineArgs { int n; int loss_dst[16]; int term_dst[16]; };
struct LossCombineKArgs { const float* pairs; float* slots; LossCombineArgs c; };
__global__ void loss_combine_kernel(const Twin<LossCombineKArgs> tw)
{
    const LossCombineKArgs& ka_ = tw.v[blockIdx.z];      // (by reference: the run-time indices c.*_dst[k] then read the argument segment, not a scratch copy)
    const float* __restrict__ pairs = ka_.pairs;
    float* __restrict__ slots = ka_.slots;
    const LossCombineArgs& c = ka_.c;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < c.n; ++k) {
        if (c.term_dst[k] >= 0) slots[c.term_dst[k]] += pairs[2 * k + 1];
        if (c.loss_dst[k] >= 0) slots[c.loss_dst[k]] += pairs[2 * k];
    }
}

// torch.optim.Adam single-tensor math on a flat buffer (weight_decay 0, amsgrad off):
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
struct AdamKArgs { float* p; float* g; float* m; float* v; long long n; AdamCoef c;
                   float* g2; int zero; };      // g2 (nullable): a second gradient buffer, added to g; zero: clear the gradient buffer(s) behind the read
__global__ void __launch_bounds__(256) adam_kernel(const Twin<AdamKArgs> tw)
{
    const AdamKArgs ka_ = tw.v[blockIdx.z];
    float* __restrict__ p = ka_.p;
    float* __restrict__ g = ka_.g;
    float* __restrict__ g2 = ka_.g2;
    const int zero = ka_.zero;
    float* __restrict__ m = ka_.m;
    float* __restrict__ v = ka_.v;
    long long n = ka_.n;
    const AdamCoef c = ka_.c;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        if (g2) { const float4 h = reinterpret_cast<const float4*>(g2)[i]; gg.x += h.x; gg.y += h.y; gg.z += h.z; gg.w += h.w; }
        if (zero) {
            reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g2) reinterpret_cast<float4*>(g2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pf = reinterpret_cast<float*>(&pp);
        float* gf = reinterpret_cast<float*>(&gg);
        float* mf = reinterpret_cast<float*>(&mm);
        float* vf = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_elem(pf[k], gf[k], mf[k], vf[k], c);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    const long long base = n4 << 2;
    const long long t = base + (long long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        const float gr = g[t] + (g2 ? g2[t] : 0.f);
        if (zero) { g[t] = 0.f; if (g2) g2[t] = 0.f; }
        float pn = p[t], mn = m[t], vn = v[t];
        adam_elem(pn, gr, mn, vn, c);
        m[t] = mn; v[t] = vn; p[t] = pn;
    }
}

struct AxpyKArgs { float* y; const float* x; float alpha; long long n; };
__global__ void axpy_kernel(const Twin<AxpyKArgs> tw)
{
    const AxpyKArgs ka_ = tw.v[blockIdx.z];
    float* __restrict__ y = ka_.y;
    const float* __restrict__ x = ka_.x;
    float alpha = ka_.alpha;
    long long n = ka_.n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

static unsigned ew_blocks(long long total, int bs)
{
    long long b = cdiv_ll(total, bs);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

int mcvc_prep_input_launch(const float* x, const float* mask, float* xin, int N, int P, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 16.0 * N * P);
    mcvc_launch(prep_input_kernel, dim3(ew_blocks((long long)N * P, 256)), dim3(256), 0, s, PrepInputKArgs{x, mask, xin, N, P});
    return (int)hipGetLastError();
}

int mcvc_mask_grad_launch(const float* dxin, const float* slabs, long long slab_stride, int nslab, const float* mask, float* dx,
                          int N, int P, int C, int accumulate, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * N * P * (nslab + 2));
    mcvc_launch(mask_grad_kernel, dim3(ew_blocks((long long)N * P, 256)), dim3(256), 0, s, MaskGradKArgs{dxin, slabs, slab_stride, nslab, mask, dx, N, P, C, accumulate});
    return (int)hipGetLastError();
}

int mcvc_bias_grad_launch(const float* dy, long long sn, long long sc, int N, int C, int P, float* db, hipStream_t s)
{
    TraceScope ts(K_BIAS_GRAD, s, 0.0, 4.0 * (double)N * C * P);
    mcvc_launch(bias_grad_kernel, dim3((unsigned)C), dim3(1024), 0, s, BiasGradKArgs{dy, sn, sc, N, P, db});
    return (int)hipGetLastError();
}

int mcvc_l1_loss_launch(const float* a, const float* b, long long n, float weight, float* loss_slot, float* term_slot,
                        float* grad_a, int accumulate, hipStream_t s)
{
    TraceScope ts(K_LOSS, s, 0.0, 12.0 * n);
    mcvc_launch(l1_loss_kernel, dim3(1), dim3(1024), 0, s, L1LossKArgs{a, b, n, weight, loss_slot, term_slot, grad_a, accumulate});
    return (int)hipGetLastError();
}

int mcvc_lsgan_loss_launch(const float* d, long long n, float target, float weight, float* loss_slot, float* term_slot,
                           float* grad_logit, hipStream_t s)
{
    TraceScope ts(K_LOSS, s, 0.0, 8.0 * n);
    mcvc_launch(lsgan_loss_kernel, dim3(1), dim3(1024), 0, s, LsganLossKArgs{d, n, target, weight, loss_slot, term_slot, grad_logit});
    return (int)hipGetLastError();
}

int mcvc_loss_combine_launch(const float* pairs, int n, const int* loss_dst, const int* term_dst, float* slots, hipStream_t s)
{
    if (n < 0 || n > 16) return (int)hipErrorInvalidValue;
    LossCombineArgs c;
    c.n = n;
    for (int k = 0; k < n; ++k) { c.loss_dst[k] = loss_dst[k]; c.term_dst[k] = term_dst[k]; }
    TraceScope ts(K_LOSS, s, 0.0, 16.0 * n);
    mcvc_launch(loss_combine_kernel, dim3(1), dim3(64), 0, s, LossCombineKArgs{pairs, slots, c});
    return (int)hipGetLastError();
}

AdamCoef mcvc_adam_coef(float lr, float b1, float b2, float eps, int step, float grad_scale)
{
    // bias corrections in double like torch.optim.Adam's python scalars, then rounded once
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    return AdamCoef{(float)((double)lr / bc1), b1, b2, eps, (float)sqrt(bc2), grad_scale};
}

int mcvc_adam_launch(float* p, float* g, float* g2, int zero_grads, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                     int step, float grad_scale, hipStream_t s)
{
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)g2 | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return MCVC_ERR_INVALID;
    TraceScope ts(K_ADAM, s, 0.0, (28.0 + (g2 ? 4.0 : 0.0) + (zero_grads ? (g2 ? 8.0 : 4.0) : 0.0)) * n);
    mcvc_launch(adam_kernel, dim3(ew_blocks(n >> 2, 256)), dim3(256), 0, s,
                AdamKArgs{p, g, m, v, n, mcvc_adam_coef(lr, b1, b2, eps, step, grad_scale), g2, zero_grads});
    return (int)hipGetLastError();
}

int mcvc_axpy_launch(float* y, const float* x, float alpha, long long n, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 12.0 * n);
    mcvc_launch(axpy_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, s, AxpyKArgs{y, x, alpha, n});
    return (int)hipGetLastError();
}
