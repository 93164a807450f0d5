// Achievable fp32 MFMA rate of this GPU: register-only v_mfma_f32_32x32x2_f32 loop (the instruction of every fp32 GEMM / conv kernel here),
// as a function of independent accumulators per wave and waves per SIMD, short and long (clocks settled) runs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_f32.hip -o tools/bin/mfma_peak_f32 && tools/bin/mfma_peak_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
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
        double flop = (double)cus * blocks_per_cu * 4 /*waves*/ * iters * 16.0 * NACC * 4096.0;
        if (rep == 2) printf("acc/wave=%d waves/SIMD=%d  %8.3f ms  %6.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flop / ms * 1e-9);
    }
    hipFree(out);
}
int main()
{
    run<1>(1, 4000); run<2>(1, 2000); run<4>(1, 1000);
    run<1>(2, 4000); run<2>(2, 2000); run<1>(3, 4000); run<1>(4, 4000);
    run<1>(2, 100000);    // ~100+ ms: long enough for the clocks to settle
    run<2>(2, 50000);
    return 0;
}
