// Do the four waves of a workgroup that read the SAME global stream (16 bytes per lane, 1 KB per wave-instruction, a few hundred cycles apart)
// share the fetch in the CU's vector L1, or does each wave pay for it in L2 bandwidth?  Every workgroup streams its own `bytes_per_wg` slice;
// `sharers` waves of it read the slice, the others idle.  Time ~ constant in `sharers` => L1 serves the repeats.
//   hipcc --offload-arch=gfx950 -O3 tools/l1_share_probe.hip -o /tmp/l1p && /tmp/l1p
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) probe(const uint4* __restrict__ src, float* out, long long pieces_per_wg, int sharers, int skew)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= sharers) return;
    const uint4* p = src + (long long)blockIdx.x * pieces_per_wg + lane;
    unsigned acc = 0;
    // (skew: wave w starts w * skew wave-instructions later in time, not in address -- a crude stagger)
    for (int i = 0; i < wave * skew; ++i) acc += __builtin_amdgcn_s_memtime() & 1;
    for (long long i = 0; i < pieces_per_wg; i += 64 * 4) {
        uint4 v0 = p[i], v1 = p[i + 64], v2 = p[i + 128], v3 = p[i + 192];
        acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
    if (acc == 0x12345u) out[blockIdx.x * 256 + threadIdx.x] = 1.f;
}
int main()
{
    const int wgs = 256 * 2;
    const long long pieces = 4096 * 16;                 // 1 MB per workgroup (L2-resident chip-wide: 512 MB total is not -- use 256 KB)
    const long long ppw = pieces / 4;                   // 256 KB per workgroup: 128 MB in all
    uint4* src; float* out;
    hipMalloc(&src, (size_t)wgs * ppw * 16); hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipMemset(src, 1, (size_t)wgs * ppw * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int skew = 0; skew <= 64; skew += 64)
        for (int sharers = 1; sharers <= 4; ++sharers) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, src, out, ppw, sharers, skew);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double gb = (double)wgs * ppw * 16 / 1e9;
            printf("skew %2d  %d wave(s) read the stream: %.3f ms  unique %.1f GB/s  requested %.1f GB/s\n", skew, sharers, best, gb / best * 1e3, gb * sharers / best * 1e3);
        }
    return 0;
}
