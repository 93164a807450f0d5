// Register-only v_mfma_f32_32x32x16_bf16 loops shaped like the conv kernel's wave tile: MT x NT accumulators, MT + NT operand fragments,
// one or two waves per SIMD.  What the matrix pipe sustains for THAT instruction mix (no LDS, no memory).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_tile_bf16.hip -o tools/bin/mfma_tile_bf16 && tools/bin/mfma_tile_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MT, int NT>
__global__ void __launch_bounds__(256) tile_loop(float* out, int iters, float seed)
{
    f32x16 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[MT], b[NT];
    for (int i = 0; i < MT; ++i) for (int k = 0; k < 8; ++k) a[i][k] = (__bf16)(seed + threadIdx.x + k + i);
    for (int j = 0; j < NT; ++j) for (int k = 0; k < 8; ++k) b[j][k] = (__bf16)(seed * 0.5f + k + j);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MT, int NT>
static void run(int blocks_per_cu, int iters)
{
    int cus = 256;
    float* out; hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((tile_loop<MT, NT>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)cus * blocks_per_cu * 4 * iters * 4.0 * MT * NT * 32768.0;
        if (rep == 2) printf("tile %d x %d accumulators  waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", MT, NT, blocks_per_cu, ms, flop / ms * 1e-9);
    }
    hipFree(out);
}
int main()
{
    run<4, 4>(1, 4000); run<2, 4>(1, 8000); run<2, 4>(2, 8000); run<2, 2>(1, 16000); run<2, 2>(2, 16000); run<2, 2>(3, 16000); run<1, 4>(2, 16000);
    return 0;
}
