// Measurement only (r6): what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for streams of EXACTLY known size, per access width?
// The guide (MI355X_MICROARCH.md, HBM) calibrates only one case -- a wide coalesced read (16 B per lane) reports half its bytes -- and calls
// every other width and WRITE_SIZE "uncalibrated".  The library's elementwise / norm / transform kernels read and write 4 bytes per lane, its
// per-plane kernels 64-byte runs: this probe gives the factor for each, so that `hbm_bytes_per_step_pmc` can be read for what it is.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cal_f -o x -- /tmp/pmc_calib ; same with WRITE_SIZE ; tools/rocpd_pmc.py <db> calib_
// Every kernel moves N floats in and N floats out (N = 64 Mi floats = 256 MiB each way: beyond the 256 MiB Infinity Cache together).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) calib_copy_b32(const float* __restrict__ a, float* __restrict__ b, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = a[i] + 1.0f;
}
__global__ void __launch_bounds__(256) calib_copy_b128(const float4* __restrict__ a, float4* __restrict__ b, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = a[i]; v.x += 1.0f; b[i] = v;
    }
}
// 16 lanes = one 64-byte run; consecutive 16-lane groups of a wave are `stride` floats apart (a per-plane kernel with 16-element planes
// that walks channels: norm_*_reg_kernel<16, 1>)
__global__ void __launch_bounds__(256) calib_copy_run64(const float* __restrict__ a, float* __restrict__ b, long long n, int stride)
{
    const long long runs = n / 16, per = stride / 16;          // a block of 4 groups x per rows tiles the buffer exactly
    for (long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); g < runs; g += (long long)gridDim.x * 16) {
        const long long blk = g / (4 * per), r = g - blk * 4 * per;
        const long long row = (r & 3) * per + (r >> 2);        // neighbouring groups of a wave -> rows `stride` floats apart
        const long long i = (blk * 4 * per + row) * 16 + (threadIdx.x & 15);
        b[i] = a[i] + 1.0f;
    }
}
__global__ void __launch_bounds__(256) calib_read_b32(const float* __restrict__ a, float* __restrict__ out, long long n)
{
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += a[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_write_b32(float* __restrict__ b, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = (float)i;
}
__global__ void __launch_bounds__(256) calib_write_b128(float4* __restrict__ b, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) b[i] = make_float4((float)i, 0.f, 1.f, 2.f);
}

int main()
{
    const long long n = 64LL << 20;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, double bytes, auto&& f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s %8.1f MB moved  %7.3f ms  %6.2f TB/s\n", name, bytes / 1e6, ms, bytes / ms / 1e9);
    };
    const unsigned G = 8192;
    run("calib_copy_b32", 8.0 * n, [&] { hipLaunchKernelGGL(calib_copy_b32, dim3(G), dim3(256), 0, 0, a, b, n); });
    run("calib_copy_b128", 8.0 * n, [&] { hipLaunchKernelGGL(calib_copy_b128, dim3(G), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });
    run("calib_copy_run64", 8.0 * n, [&] { hipLaunchKernelGGL(calib_copy_run64, dim3(G), dim3(256), 0, 0, a, b, n, 4096); });
    run("calib_read_b32", 4.0 * n, [&] { hipLaunchKernelGGL(calib_read_b32, dim3(G), dim3(256), 0, 0, a, b, n); });
    run("calib_write_b32", 4.0 * n, [&] { hipLaunchKernelGGL(calib_write_b32, dim3(G), dim3(256), 0, 0, b, n); });
    run("calib_write_b128", 4.0 * n, [&] { hipLaunchKernelGGL(calib_write_b128, dim3(G), dim3(256), 0, 0, (float4*)b, n / 4); });
    return 0;
}
