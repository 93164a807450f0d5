// Achievable bf16 MFMA rate of this GPU: register-only v_mfma_f32_32x32x16_bf16 loop (the instruction of the bf16 inference path).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o tools/bin/mfma_peak_bf16 && tools/bin/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(seed + threadIdx.x + k); b[k] = (__bf16)(seed * 0.5f + k); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(int blocks_per_cu, int iters)
{
    int cus = 256;
    float* out; hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)cus * blocks_per_cu * 4 /*waves*/ * iters * 8.0 * NACC * 32768.0;
        if (rep == 2) printf("acc/wave=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flop / ms * 1e-9);
    }
    hipFree(out);
}
int main()
{
    run<1>(1, 20000); run<2>(1, 10000); run<4>(1, 5000); run<4>(2, 5000); run<2>(2, 10000);
    run<4>(2, 50000);     // ~100 ms: long enough for the clocks to settle
    return 0;
}
