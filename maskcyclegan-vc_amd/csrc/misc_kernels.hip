// Small HBM-bound kernels of the training step: FIF input conditioning, loss reductions with their
// gradients, bias gradients, flat Adam, axpy.
//
// Reference call sites: model.py:241 (stack(x*mask, mask)); train.py:219-237 and :276-294 (L1 and
// LSGAN means); train.py:119-122, 242, 299 (torch.optim.Adam, betas (0.5, 0.999), eps 1e-8).
#include "mcvc_common.h"
#include "misc.h"
#include "trace.h"

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
__global__ void prep_input_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ xin, int N, int P)
{
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
__global__ void mask_grad_kernel(const float* __restrict__ dxin, const float* __restrict__ dxin_slabs, long long slab_stride, int nslab,
                                 const float* __restrict__ mask, float* __restrict__ dx, int N, int P, int C, int accumulate)
{
    const long long total = (long long)N * P;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / P;
        const int i = (int)(idx - n * P);
        const long long so = (n * C) * P + i;
        float v = dxin[so];
        for (int sl = 1; sl < nslab; ++sl) v += dxin_slabs[(long long)(sl - 1) * slab_stride + so];
        if (mask) v *= mask[idx];
        dx[idx] = accumulate ? dx[idx] + v : v;
    }
}

// db[c] += sum_{n,i} dy[n][c][i]      (one block per channel)
// db[c] += sum over (n, pixel) of dy: one workgroup of 1024 threads per channel, 16-byte loads, four independent partial sums per thread
// (a 256-thread scalar loop took 200-500 us per launch at 32-64 samples: the Cout = 1 layers reduce 330 k elements in ONE workgroup).
// Fixed summation order: deterministic.
__global__ void __launch_bounds__(1024) bias_grad_kernel(const float* __restrict__ dy, long long sn, long long sc, int N, int P, float* __restrict__ db)
{
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
__global__ void __launch_bounds__(1024) l1_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float weight,
                                                       float* __restrict__ loss_slot, float* __restrict__ term_slot,
                                                       float* __restrict__ grad_a, int accumulate)
{
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
__global__ void __launch_bounds__(1024) lsgan_loss_kernel(const float* __restrict__ d, long long n, float target, float weight,
                                                          float* __restrict__ loss_slot, float* __restrict__ term_slot,
                                                          float* __restrict__ grad_logit)
{
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
__global__ void loss_combine_kernel(const float* __restrict__ pairs, float* __restrict__ slots, const LossCombineArgs c)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < c.n; ++k) {
        if (c.term_dst[k] >= 0) slots[c.term_dst[k]] += pairs[2 * k + 1];
        if (c.loss_dst[k] >= 0) slots[c.loss_dst[k]] += pairs[2 * k];
    }
}

// torch.optim.Adam single-tensor math on a flat buffer (weight_decay 0, amsgrad off):
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float sqrt_bc2, float grad_scale)
{
    const float step_size = lr / bc1;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pf = reinterpret_cast<float*>(&pp);
        float* gf = reinterpret_cast<float*>(&gg);
        float* mf = reinterpret_cast<float*>(&mm);
        float* vf = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = gf[k] * grad_scale;
            mf[k] = mf[k] + (gr - mf[k]) * (1.0f - b1);
            vf[k] = vf[k] * b2 + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(vf[k]) / sqrt_bc2 + eps;
            pf[k] = pf[k] - step_size * (mf[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    const long long base = n4 << 2;
    const long long t = base + (long long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        const float gr = g[t] * grad_scale;
        const float mn = m[t] + (gr - m[t]) * (1.0f - b1);
        const float vn = v[t] * b2 + (1.0f - b2) * gr * gr;
        m[t] = mn; v[t] = vn;
        p[t] = p[t] - step_size * (mn / (sqrtf(vn) / sqrt_bc2 + eps));
    }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long long n)
{
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
    hipLaunchKernelGGL(prep_input_kernel, dim3(ew_blocks((long long)N * P, 256)), dim3(256), 0, s, x, mask, xin, N, P);
    return (int)hipGetLastError();
}

int mcvc_mask_grad_launch(const float* dxin, const float* slabs, long long slab_stride, int nslab, const float* mask, float* dx,
                          int N, int P, int C, int accumulate, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * N * P * (nslab + 2));
    hipLaunchKernelGGL(mask_grad_kernel, dim3(ew_blocks((long long)N * P, 256)), dim3(256), 0, s, dxin, slabs, slab_stride, nslab, mask, dx, N, P, C, accumulate);
    return (int)hipGetLastError();
}

int mcvc_bias_grad_launch(const float* dy, long long sn, long long sc, int N, int C, int P, float* db, hipStream_t s)
{
    TraceScope ts(K_BIAS_GRAD, s, 0.0, 4.0 * (double)N * C * P);
    hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)C), dim3(1024), 0, s, dy, sn, sc, N, P, db);
    return (int)hipGetLastError();
}

int mcvc_l1_loss_launch(const float* a, const float* b, long long n, float weight, float* loss_slot, float* term_slot,
                        float* grad_a, int accumulate, hipStream_t s)
{
    TraceScope ts(K_LOSS, s, 0.0, 12.0 * n);
    hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(1024), 0, s, a, b, n, weight, loss_slot, term_slot, grad_a, accumulate);
    return (int)hipGetLastError();
}

int mcvc_lsgan_loss_launch(const float* d, long long n, float target, float weight, float* loss_slot, float* term_slot,
                           float* grad_logit, hipStream_t s)
{
    TraceScope ts(K_LOSS, s, 0.0, 8.0 * n);
    hipLaunchKernelGGL(lsgan_loss_kernel, dim3(1), dim3(1024), 0, s, d, n, target, weight, loss_slot, term_slot, grad_logit);
    return (int)hipGetLastError();
}

int mcvc_loss_combine_launch(const float* pairs, int n, const int* loss_dst, const int* term_dst, float* slots, hipStream_t s)
{
    if (n < 0 || n > 16) return (int)hipErrorInvalidValue;
    LossCombineArgs c;
    c.n = n;
    for (int k = 0; k < n; ++k) { c.loss_dst[k] = loss_dst[k]; c.term_dst[k] = term_dst[k]; }
    TraceScope ts(K_LOSS, s, 0.0, 16.0 * n);
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, s, pairs, slots, c);
    return (int)hipGetLastError();
}

int mcvc_adam_launch(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                     int step, float grad_scale, hipStream_t s)
{
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return MCVC_ERR_INVALID;
    // bias corrections in double like torch.optim.Adam's python scalars, then rounded once
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    TraceScope ts(K_ADAM, s, 0.0, 28.0 * n);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n >> 2, 256)), dim3(256), 0, s, p, g, m, v, n, (float)((double)lr / bc1), b1, b2, eps, 1.0f, (float)sqrt(bc2), grad_scale);
    return (int)hipGetLastError();
}

int mcvc_axpy_launch(float* y, const float* x, float alpha, long long n, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 12.0 * n);
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, s, y, x, alpha, n);
    return (int)hipGetLastError();
}
