// Measurement only: cost of an in-kernel all-to-all hand-off (every workgroup writes a slice, then reads everybody's slices)
// as a function of WHERE the workgroups sit and of the cache policy of the stores / loads / counters.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_handoff.hip -o tools/bin/xcd_handoff && tools/bin/xcd_handoff
// Modes (store | load | counter scope | placement):
//   0  sc1 store | sc1 load        | agent     | 64 WGs wherever the dispatcher puts them (= the round-2 persistent trunk)
//   1  sc1 store | sc1 load        | agent     | all on one XCD
//   2  plain     | sc0 load        | workgroup | one XCD
//   3  plain     | sc0|sc1 load    | workgroup | one XCD
//   4  plain     | sc1 load        | workgroup | one XCD
//   5  plain     | buffer_inv sc1 + plain load | workgroup | one XCD
//   6  plain     | plain load (expected WRONG: stale L1) | workgroup | one XCD
//   7  plain     | sc1 load        | agent     | one XCD
// Every round is verified (wrong values are counted), so a fast but incoherent policy shows up as errors, not as a win.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 512;
typedef int v4i_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float expect(int r, int i) { return (float)((r * 131 + i * 7) & 0xFFFF); }

template <int MODE>
__global__ void __launch_bounds__(kThreads) handoff(float* buf, unsigned* sync, unsigned* out, int rounds, int nwg, int floats, int target_xcd)
{
    constexpr bool placed = MODE >= 1;
    constexpr bool agent_ctr = (MODE == 0 || MODE == 1 || MODE == 7);
    __shared__ int s_id;
    __shared__ float s_ok;
    const int tid = threadIdx.x;
    int id = blockIdx.x;
    if (placed) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;      // HW_REG_XCC_ID[3:0]
        if ((int)xcc != target_xcd) return;
        if (tid == 0) s_id = (int)__hip_atomic_fetch_add(sync + 1023, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        id = s_id;
        if (id >= nwg) return;
        if (tid == 0) atomicAdd(out + 2 + (blockIdx.x & 7), 1u);                     // which launch slots landed on the target
    }
    const int per = floats / nwg;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        float* b = buf + (size_t)(r & 1) * floats;
        for (int i = tid; i < per; i += kThreads) {
            const int e = id * per + i;
            if (MODE == 0 || MODE == 1) __hip_atomic_store(b + e, expect(r, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else b[e] = expect(r, e);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (agent_ctr) __hip_atomic_fetch_add(sync + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(sync + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool ok = false;
            for (unsigned spins = 0; spins < (1u << 20); ++spins) {
                const unsigned v = agent_ctr ? __hip_atomic_load(sync + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_fetch_add(sync + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v >= (unsigned)nwg) { ok = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            s_ok = ok ? 1.f : 0.f;
        }
        __syncthreads();
        if (s_ok == 0.f) { if (tid == 0) atomicAdd(out + 1, 1u); return; }
        if (MODE == 5) asm volatile("buffer_inv sc1" ::: "memory");
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(b, 0, floats * 4, 0x00020000);
        for (int f = tid; f < floats / 4; f += kThreads) {
            float4 v;
            if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 7) v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, f * 16, 0, 16));
            else if (MODE == 2) v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, f * 16, 0, 1));
            else if (MODE == 3) v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, f * 16, 0, 17));
            else v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, f * 16, 0, 0));
            bad += (v.x != expect(r, 4 * f)) + (v.y != expect(r, 4 * f + 1)) + (v.z != expect(r, 4 * f + 2)) + (v.w != expect(r, 4 * f + 3));
        }
        __syncthreads();
    }
    if (bad) atomicAdd(out, bad);
}

template <int MODE>
static void run(int nwg, int floats, int rounds, int xcd)
{
    float* buf; unsigned *sync, *out;
    hipMalloc(&buf, sizeof(float) * 2 * floats); hipMalloc(&sync, 4096); hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = MODE >= 1 ? 8 * nwg : nwg;
    float best = 1e9f; unsigned h[10] = {0};
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(buf, 0xFF, sizeof(float) * 2 * floats); hipMemset(sync, 0, 4096); hipMemset(out, 0, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(handoff<MODE>, dim3(grid), dim3(kThreads), 0, 0, buf, sync, out, rounds, nwg, floats, xcd);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
    }
    printf("mode %d  wgs %3d  %6d floats  xcd %d : %7.2f us/round  wrong=%u timeouts=%u  slots=[%u %u %u %u %u %u %u %u]\n", MODE, nwg, floats, xcd,
           1e3f * best / rounds, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
    hipFree(buf); hipFree(sync); hipFree(out);
}

int main(int argc, char** argv)
{
    const int rounds = 400;
    for (int floats : {8192, 16384}) {
        for (int nwg : {32, 64}) {
            run<0>(nwg, floats, rounds, 0);
            run<1>(nwg, floats, rounds, 0);
            run<2>(nwg, floats, rounds, 0);
            run<3>(nwg, floats, rounds, 0);
            run<4>(nwg, floats, rounds, 0);
            run<5>(nwg, floats, rounds, 0);
            run<6>(nwg, floats, rounds, 0);
            run<7>(nwg, floats, rounds, 3);
        }
    }
    return 0;
}
