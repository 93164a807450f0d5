#pragma once
#include <hip/hip_runtime.h>

int mcvc_prep_input_launch(const float* x, const float* mask, float* xin, int N, int P, hipStream_t s);
int mcvc_mask_grad_launch(const float* dxin, const float* slabs, long long slab_stride, int nslab, const float* mask, float* dx,
                          int N, int P, int C, int accumulate, hipStream_t s);
int mcvc_bias_grad_launch(const float* dy, long long sn, long long sc, int N, int C, int P, float* db, hipStream_t s);
int mcvc_l1_loss_launch(const float* a, const float* b, long long n, float weight, float* loss_slot, float* term_slot,
                        float* grad_a, int accumulate, hipStream_t s);
int mcvc_lsgan_loss_launch(const float* d, long long n, float target, float weight, float* loss_slot, float* term_slot,
                           float* grad_logit, hipStream_t s);
int mcvc_loss_combine_launch(const float* pairs, int n, const int* loss_dst, const int* term_dst, float* slots, hipStream_t s);
int mcvc_adam_launch(float* p, float* g, float* g2, int zero_grads, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                     int step, float grad_scale, hipStream_t s);
int mcvc_axpy_launch(float* y, const float* x, float alpha, long long n, hipStream_t s);

// torch.optim.Adam's per-element update (weight_decay 0, amsgrad off); shared by adam_kernel (misc_kernels.hip) and the optimizer step
// fused with the weight re-pack (update_net_kernel, pack_kernels.hip) so that both produce the same bits.
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g*g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps),  step_size = lr / bc1 (rounded once, host)
struct AdamCoef { float step_size, b1, b2, eps, sqrt_bc2, grad_scale; };
static __device__ __forceinline__ void adam_elem(float& p, float g_raw, float& m, float& v, const AdamCoef& c)
{
    const float gr = g_raw * c.grad_scale;
    m = m + (gr - m) * (1.0f - c.b1);
    v = v * c.b2 + (1.0f - c.b2) * gr * gr;
    const float denom = sqrtf(v) / c.sqrt_bc2 + c.eps;
    p = p - c.step_size * (m / denom);
}
// host: the coefficients of step `step` (bias corrections in double like torch.optim.Adam's python scalars, then rounded once)
AdamCoef mcvc_adam_coef(float lr, float b1, float b2, float eps, int step, float grad_scale);
