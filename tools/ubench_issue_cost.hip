// Micro-benchmark (gfx950): what one memory instruction costs a wave that is otherwise issuing MFMAs back to back, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue_cost.hip -o /tmp/ubench_issue_cost && /tmp/ubench_issue_cost
// Body = 16 independent v_mfma_f32_32x32x16_bf16 (512 cycles per SIMD at the 32-cycle issue rate) with P memory operations spread between
// them:  mode 0 nothing, 1 LDS-DMA (global_load_lds_dwordx4, 1 KB per wave), 2 global_load_dwordx4 into registers + ds_write_b128 of the
// previous body's registers, 3 the loads only, 4 the LDS stores only, 5 ds_read_b128.
// Output: cycles per body (s_memtime) per mode / P, and the extra cycles per memory instruction over mode 0.
// (decides between LDS-DMA and register staging for streamed weights.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int P, int R = 0>
__global__ void __launch_bounds__(256) body_kernel(const u32x4* __restrict__ src, float* __restrict__ out, long long* __restrict__ cyc, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 seed = src[tid];
    bf16x8 a = __builtin_bit_cast(bf16x8, seed), b = __builtin_bit_cast(bf16x8, src[tid + 256]);
    const u32x4* p = src + (size_t)blockIdx.x * 4096 + tid;                     // 64 KB per block, L2-resident after the first pass
    u32x4 rc[P > 0 ? P : 1], rn[P > 0 ? P : 1];
#pragma unroll
    for (int i = 0; i < (P > 0 ? P : 1); ++i) { rc[i] = seed; rn[i] = seed; }
    u32x4 rd = seed;
    u32x4 rr[2][R > 0 ? R : 1];
#pragma unroll
    for (int i = 0; i < (R > 0 ? R : 1); ++i) { rr[0][i] = seed; rr[1][i] = seed; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    // one body: 16 MFMAs with P memory operations between them; loads go into `ld`, LDS stores / consumers use `st` (the registers the
    // PREVIOUS body loaded: two register sets alternate, no copies, so no instruction waits for a load of its own body)
    auto body = [&](int it, u32x4 (&ld)[P > 0 ? P : 1], u32x4 (&st)[P > 0 ? P : 1]) __attribute__((always_inline)) {
        constexpr int GAP = P > 0 ? 16 / P : 16;
        if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < P; ++i) rd ^= st[i];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 7], 0, 0, 0);
            if (P > 0 && j % GAP == GAP / 2) {
                const int i = j / GAP;
                const int slot = ((it * P + i) & 15) * 256;
                if (MODE == 1)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + slot),
                                                     (__attribute__((address_space(3))) void*)(smem + (size_t)(i * 256 + wave * 64) * 16), 16, 0, 0);
                if (MODE == 2 || MODE == 3) ld[i] = p[slot];
                if (MODE == 2 || MODE == 4) *reinterpret_cast<u32x4*>(smem + (size_t)(i * 256 + tid) * 16) = st[i];
                if (MODE == 5) ld[i] = *reinterpret_cast<const u32x4*>(smem + (size_t)(i * 256 + tid) * 16);
            }
        }
        // pin the interleave (left alone the scheduler gathers the memory instructions at the head of the body)
        if (P > 0) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
                if (MODE == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                if (MODE == 2 || MODE == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (MODE == 2 || MODE == 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (MODE == 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P > 0 ? P : 0) : "memory");     // previous body's pieces landed (keeps the queue bounded)
        __builtin_amdgcn_sched_barrier(0);
        if (R > 0) {       // R operand reads at the head of the next body (as the conv kernel issues them), consumed one body later
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < R; ++i) rd ^= rr[it & 1][i];
#pragma unroll
            for (int i = 0; i < R; ++i)
                asm volatile("ds_read_b128 %0, %1" : "=&v"(rr[it & 1][i]) : "v"((unsigned)(((i * 256 + tid) * 16 + 65536))));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it < iters; it += 8) {          // (8 bodies per trip: the loop head's conservative vmcnt(0) is paid once per 8)
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            body(it + u, rn, rc);
            body(it + u + 1, rc, rn);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    s += (float)(rc[0].x ^ rd.x) + (float)smem[tid * 16];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int P, int R = 0>
double run(const u32x4* src, float* out, long long* cyc, int iters, double* ms_out)
{
    auto kern = body_kernel<MODE, P, R>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 128 * 1024, 0, src, out, cyc, 200);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 128 * 1024, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (long long v : h) avg += (double)v;
    *ms_out = ms;
    return avg / 256.0 / iters;                                                   // s_memtime ticks (100 MHz constant clock on gfx9) per body
}

int main()
{
    const size_t n = (size_t)256 * 4096 + 4096;
    u32x4* src; float* out; long long* cyc;
    hipMalloc(&src, n * sizeof(u32x4)); hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(long long));
    hipMemset(src, 0x3c, n * sizeof(u32x4));
    const int iters = 4000;
    double ms0 = 0, ms = 0;
    run<0, 0>(src, out, cyc, iters, &ms0);
    const double ticks0 = run<0, 0>(src, out, cyc, iters, &ms0);
    const double us_body0 = ms0 * 1e3 / iters;
    printf("mode 0: %.1f counter ticks per body (s_memtime / readcyclecounter) -> %.3f ticks per ns\n", ticks0, ticks0 / (us_body0 * 1e3));
    printf("mode 0 (16 MFMAs only): %.4f us per body  (%.1f cycles at 2.4 GHz, %.1f at 1.9 GHz)\n", us_body0, us_body0 * 2400, us_body0 * 1900);
#define RUN(M, P_, name)                                                                                                                   \
    {                                                                                                                                      \
        run<M, P_>(src, out, cyc, iters, &ms);                                                                                             \
        const double us = ms * 1e3 / iters;                                                                                                \
        printf("mode %d %-38s P=%d: %.4f us per body, +%.1f ns per instruction (= %.0f cycles at 1.9 GHz)\n", M, name, P_, us,         \
               (us - us_body0) * 1e3 / P_, (us - us_body0) * 1e3 / P_ * 1.9);                                                              \
    }
    RUN(1, 1, "LDS-DMA") RUN(1, 2, "LDS-DMA") RUN(1, 4, "LDS-DMA") RUN(1, 8, "LDS-DMA")
    RUN(3, 1, "global_load_dwordx4") RUN(3, 2, "global_load_dwordx4") RUN(3, 4, "global_load_dwordx4") RUN(3, 8, "global_load_dwordx4")
    RUN(4, 1, "ds_write_b128") RUN(4, 2, "ds_write_b128") RUN(4, 4, "ds_write_b128") RUN(4, 8, "ds_write_b128")
    RUN(2, 1, "load + ds_write (pair)") RUN(2, 2, "load + ds_write (pair)") RUN(2, 4, "load + ds_write (pair)") RUN(2, 8, "load + ds_write (pair)")
    RUN(5, 2, "ds_read_b128") RUN(5, 8, "ds_read_b128") RUN(5, 16, "ds_read_b128")
    // the same with 8 operand reads (ds_read_b128) per body, as in the conv kernel's steps
    double ms8 = 0;
    const double tk8 = run<0, 0, 8>(src, out, cyc, iters, &ms8);
    const double us8 = ms8 * 1e3 / iters;
    printf("mode 0 + 8 ds_read_b128 per body: %.4f us per body (%.0f ticks)\n", us8, tk8);
#define RUN8(M, P_, name)                                                                                                                  \
    {                                                                                                                                      \
        const double tk = run<M, P_, 8>(src, out, cyc, iters, &ms);                                                                        \
        const double us = ms * 1e3 / iters;                                                                                                \
        printf("mode %d %-30s + 8 reads P=%d: %.4f us per body (%.0f ticks), +%.1f ns per instruction (= %.0f cycles at 1.9 GHz)\n", M, name, P_, us, tk, \
               (us - us8) * 1e3 / P_, (us - us8) * 1e3 / P_ * 1.9);                                                                        \
    }
    RUN8(1, 1, "LDS-DMA") RUN8(1, 2, "LDS-DMA") RUN8(1, 4, "LDS-DMA")
    RUN8(2, 1, "load + ds_write") RUN8(2, 2, "load + ds_write") RUN8(2, 4, "load + ds_write")
    RUN8(3, 2, "global_load_dwordx4") RUN8(4, 2, "ds_write_b128")
    return 0;
}
