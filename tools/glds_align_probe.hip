// Does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept a global address that is only 4-byte aligned?
// (implicit-GEMM operand gathers of the stride-2 3x3 layers read 4 consecutive floats starting at odd column offsets)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(const float* src, float* dst, int shift)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    const float* g = src + shift + 4 * threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
}
int main()
{
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096); hipMalloc(&d, 1024);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(d, 0, 1024);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(256);
        hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) if (o[i] != (float)(i + shift)) ++bad;
        printf("shift %d floats: err=%d mismatches=%d  first: %g %g %g %g %g\n", shift, (int)e, bad, o[0], o[1], o[2], o[3], o[4]);
    }
    return 0;
}
