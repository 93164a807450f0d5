// Measurement only: how fast can a workgroup-per-panel kernel stream a weight set that is read exactly ONCE (no reuse), by LDS-DMA
// (global_load_lds b128, ring of stages, explicit vmcnt) and by ordinary register loads?  This is the floor of the small-N Winograd products.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/bin/stream_probe && tools/bin/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

// every workgroup streams `per_wg` floats (contiguous) in stages of STAGE floats; PER = DMA instructions per thread and stage
template <int PER, int ST>
__global__ void __launch_bounds__(256) dma_stream(const float* __restrict__ src, long long per_wg, float* out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = PER * 256 * 4;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* p = src + (long long)blockIdx.x * per_wg + tid * 4;
    const int nst = (int)(per_wg / STAGE);
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) glds16(p + (long long)s * STAGE + i * 1024, smem + buf * STAGE + (wave * 64 + i * 256) * 4);
    };
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) if (s < nst) issue(s, s);
    float acc = 0.f;
    for (int st = 0; st < nst; ++st) {
        const int newer = (nst - 1 - st) < (ST - 2) ? (nst - 1 - st) : (ST - 2);
        if (newer >= 2) wait_vm<2 * PER>(); else if (newer == 1) wait_vm<PER>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + ST - 1 < nst) issue(st + ST - 1, (st + ST - 1) % ST);
        acc += smem[(st % ST) * STAGE + tid];
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int UNROLL>
__global__ void __launch_bounds__(256) reg_stream(const float4* __restrict__ src, long long per_wg4, float* out)
{
    const float4* p = src + (long long)blockIdx.x * per_wg4 + threadIdx.x;
    float acc = 0.f;
    for (long long i = 0; i + 256 * UNROLL <= per_wg4; i += 256 * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <class F>
static float timeit(F&& f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main()
{
    float* out; hipMalloc(&out, 64);
    for (long long mb : {36LL, 288LL, 1152LL}) {
        const long long floats = mb * 1024 * 1024 / 4;
        float* src; hipMalloc(&src, floats * 4); hipMemset(src, 0, floats * 4);
        for (int wgs : {256, 512, 1024, 2048}) {
            const long long per = floats / wgs;
            hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const float t0 = timeit([&] { hipLaunchKernelGGL((dma_stream<2, 4>), dim3(wgs), dim3(256), 4 * 2 * 4096, 0, src, per, out); });
            const float t1 = timeit([&] { hipLaunchKernelGGL((dma_stream<4, 4>), dim3(wgs), dim3(256), 4 * 4 * 4096, 0, src, per, out); });
            const float t2 = timeit([&] { hipLaunchKernelGGL((dma_stream<4, 8>), dim3(wgs), dim3(256), 8 * 4 * 4096, 0, src, per, out); });
            const float t3 = timeit([&] { hipLaunchKernelGGL((reg_stream<4>), dim3(wgs), dim3(256), 0, 0, (const float4*)src, per / 4, out); });
            const float t4 = timeit([&] { hipLaunchKernelGGL((reg_stream<8>), dim3(wgs), dim3(256), 0, 0, (const float4*)src, per / 4, out); });
            const double gb = floats * 4 / 1e9;
            printf("%5lld MB %5d WGs: dma 8KBx4 %6.1f us %5.2f TB/s | dma 16KBx4 %6.1f us %5.2f | dma 16KBx8 %6.1f us %5.2f | regs x4 %6.1f us %5.2f | regs x8 %6.1f us %5.2f\n",
                   mb, wgs, t0 * 1e3, gb / t0, t1 * 1e3, gb / t1, t2 * 1e3, gb / t2, t3 * 1e3, gb / t3, t4 * 1e3, gb / t4);
        }
        hipFree(src);
    }
    return 0;
}
