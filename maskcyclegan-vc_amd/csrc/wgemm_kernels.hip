// Implicit weight gradient of the GEMM-shaped convolutions (sgemm.h, r5): dW[co][ci][tap] (+)= sum over pixels of dY[co][pixel] * x_tap[ci][pixel],
// with BOTH operands read where they lie -- dY as the InstanceNorm backward left it, x as the phase-split padded activation the forward's
// implicit GEMM gathers from -- no transposed copies (im2col_s2_t / planes_t of r2-r4), no tap planes.
//
// The contraction runs over the PIXELS, and pixels are the contiguous axis of both tensors: a stage of 32 pixels of one row (channel) is 128
// consecutive bytes.  The staged path transposed both operands to pixel-major so that the GEMM pipeline could read K-major LDS tiles; this
// kernel keeps the rows as they are ([channel][32 pixels] tiles, filled by the same 16-byte LDS-DMA pieces as every other GEMM here) and
// moves the transposition into the operand READ, where it is free: an MFMA's k index is only a label, so the two k slots of
// v_mfma_f32_32x32x2_f32 are fed from pixel 16*half + j of the stage (j = 0..15 over the 16 MFMAs of a stage) -- every lane then needs 16
// CONSECUTIVE floats of its row = four ds_read_b128.  Rows are 8 pieces of 16 bytes; piece p of row r is stored at slot p ^ ((r >> 1) & 7),
// which makes the sixteen lanes of each ds_read_b128 service group (MI355X_MICROARCH.md, LDS) hit sixteen distinct 16-byte bank slots.  The
// swizzle costs nothing either: LDS-DMA writes lane L of a wave to the L-th slot, so lane L simply FETCHES the piece that belongs there.
//
// A workgroup owns 128 output channels x CIB input channels x ALL taps: the dY tile is staged once per stage and multiplied with every tap's
// window of x (9 windows of the four phase planes for the 3 x 3 stride-2 layers), 9 x CIB / 32 accumulators per wave -- 3.5x fewer LDS-DMA
// bytes per FLOP than the 64 x 64 tiles of the staged product, and the epilogue writes the taps of one (co, ci) filter side by side: rows of
// CIB * TAPS consecutive floats of the OIHW gradient, no slab-shaped scatter.  K split over the pixels into slabs (dw_accum sums them) when
// the tile count cannot fill the chip; no atomics in any mode.
#include "sgemm.h"
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }

__device__ __forceinline__ void glds16(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

constexpr int WBM = 128, WGK = 32, WST = 3;

template <int TAPS, int CIB>
__global__ void __launch_bounds__(256) wgemm_kernel(const Twin<WGemmArgs> tw)
{
    constexpr int NJ = CIB / 32;                       // 32-channel sub-tiles of B per tap
    constexpr int NACC = TAPS * NJ;                    // accumulators per wave
    constexpr int SA = WBM * WGK, SB1 = 32 * WGK;      // floats: the dY tile, one B sub-tile
    constexpr int STAGE = SA + NACC * SB1;
    constexpr int ND = 4 + NACC;                       // LDS-DMA instructions per wave and stage
    const WGemmArgs& a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // Tile ids run input-channel tile fastest, then output-channel tile, then pixel split.  The hardware deals consecutive workgroup ids
    // round-robin to the 8 XCDs, so with lid = blockIdx.x the nt workgroups that multiply the SAME dY tile sat behind up to 8 different L2s and
    // dY was fetched once per XCD.  With one contiguous range of ids per XCD (r6; the remap of gemm2_kernel / igemm_kernel) an XCD's workgroups
    // are neighbours in (n, m) of the same pixel split -- and the splits partition the pixels, so the XCDs share almost nothing
    int lid = (int)blockIdx.x;
    if (a.xcd_order) {
        const int total = (int)gridDim.x, q = total >> 3, r = total & 7, xcd = lid & 7, k = lid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int nt = a.nt, mt = a.mt;
    const int n0 = (lid % nt) * CIB; lid /= nt;        // first input channel
    const int m0 = (lid % mt) * WBM;                   // first output channel
    const int ks = lid / mt;
    const int nst_all = a.nstages;
    const int per = (nst_all + a.nsplit - 1) / a.nsplit;
    const int st0 = ks * per;
    const int nst = (st0 + per <= nst_all ? per : (nst_all > st0 ? nst_all - st0 : 0));
    // ---- this lane's piece of every row it fills: rows r8 + 32 i of the dY tile and of every B sub-tile, slot tid & 7 -> logical piece p
    const int r8 = tid >> 3;
    const int p = (tid & 7) ^ ((r8 >> 1) & 7);
    const int OW = a.OW, P = a.P, NPIX = a.NPIX;
    const long long a_cs = a.a_cs, b_cs = a.b_cs;
    const float* const Ap = a.a + (long long)(m0 + r8) * a_cs;
    const float* const Bp = a.b + (long long)(n0 + r8) * b_cs;
    auto issue = [&](int s, int buf) {
        float* base = smem + buf * STAGE;
        const int n = (st0 + s) * WGK + 4 * p;
        const int nn = n < NPIX ? n : 0;               // (pixels beyond the last: any valid piece -- the dY operand is zeroed at the read)
        const int bb = nn / P, rem = nn - bb * P;
        const int ii = rem / OW, jj = rem - ii * OW;
        const long long offA = (long long)bb * a.a_sn + (long long)ii * a.a_pitch + jj;
        const long long offB = (long long)bb * a.b_sn + (long long)ii * a.b_pitch + jj;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16(Ap + (long long)(32 * i) * a_cs + offA, base + (wave * 64 + i * 256) * 4);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                glds16(Bp + a.boff[t] + (long long)(32 * j) * b_cs + offB, base + SA + (t * NJ + j) * SB1 + wave * 64 * 4);
    };
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < WST - 1; ++s)
        if (s < nst) issue(s, s);
    const int sw = (l31 >> 1) & 7;
    const int a_row = (wave * 32 + l31) * WGK, b_row = SA + l31 * WGK;
    for (int st = 0; st < nst; ++st) {
        const int newer = (nst - 1 - st) < (WST - 2) ? (nst - 1 - st) : (WST - 2);
        if (newer >= 1) wait_vm<ND>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + WST - 1 < nst) issue(st + WST - 1, (st + WST - 1) % WST);
        // Operand reads are inline assembly with an explicit lgkmcnt wait per accumulator group.  As plain C++ loads hipcc cannot tell that the
        // LDS-DMA just issued (stage st + 2's buffer) does not write what a ds_read_b128 of stage st reads, and put `s_waitcnt vmcnt(0)` in front
        // of the stage's first read: every stage then waited for the copy issued a few instructions earlier and the three-stage ring never
        // overlapped anything (found in the ISA in r5; the same had been fixed in igemm_kernel<AROW> and the bf16 convolution).
        const float* sb = smem + (st % WST) * STAGE;
        const unsigned sb_addr = lds_addr(sb);
        unsigned slot[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[q] = (unsigned)(((4 * half + q) ^ sw) << 4);
        f32x4 av[4], bvv[2][4];
        {
            const unsigned aa = sb_addr + (unsigned)a_row * 4u;
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1" : "=&v"(av[q]) : "v"(aa + slot[q]));
        }
        auto request_b = [&](int t, f32x4 (&b_)[4]) __attribute__((always_inline)) {
            const unsigned ba = sb_addr + (unsigned)(b_row + t * SB1) * 4u;
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1" : "=&v"(b_[q]) : "v"(ba + slot[q]));
        };
        request_b(0, bvv[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bvv[0][0]), "+v"(bvv[0][1]), "+v"(bvv[0][2]), "+v"(bvv[0][3]));
        const int k_first = (st0 + st) * WGK + 16 * half;          // the first of this lane's 16 pixels
        if (k_first + 16 > NPIX) {                                 // the last stage's tail: pixels that do not exist contribute nothing
#pragma unroll
            for (int j = 0; j < 16; ++j) if (k_first + j >= NPIX) av[j >> 2][j & 3] = 0.f;
        }
        // 1-D convolutions over DENSE rows of zw columns (the trunk): the window of tap 0 / tap 2 is shifted by -1 / +1 column and reads the
        // neighbouring row's element at a row's first / last column, where the padding holds a zero -- those x values are dropped here
        unsigned zfirst = 0, zlast = 0;                            // bit j: pixel j of this lane's 16 is a row's first / last column
        if (TAPS == 3 && a.zw) {
            int w = k_first % a.zw;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                zfirst |= (w == 0 ? 1u : 0u) << j;
                zlast |= (w == a.zw - 1 ? 1u : 0u) << j;
                w = (w + 1 == a.zw) ? 0 : w + 1;
            }
        }
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            f32x4 (&bv)[4] = bvv[t & 1];
            if (t > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
            if (t + 1 < NACC) request_b(t + 1, bvv[(t + 1) & 1]);            // the next group's window, in flight during this group's 16 MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (TAPS == 3 && a.zw && (t / NJ) != 1) {
                const unsigned zm = (t / NJ) == 0 ? zfirst : zlast;
#pragma unroll
                for (int j = 0; j < 16; ++j) if ((zm >> j) & 1u) bv[j >> 2][j & 3] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][0], bv[q][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][1], bv[q][1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][2], bv[q][2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][3], bv[q][3], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: the TAPS values of a filter (co, ci) are consecutive floats of the OIHW gradient
    const long long ldc = a.ldc;
    const int mb = m0 + wave * 32 + 4 * half;
    float* dst;
    if (ks) dst = a.c_slab + (long long)(ks - 1) * a.c_split + (long long)mb * ldc;
    else dst = (mb < a.m_split) ? a.c + (long long)mb * ldc : a.c2 + (long long)(mb - a.m_split) * ldc;
    const bool accumulate = a.accumulate && ks == 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* row = dst + (long long)((r & 3) + 8 * (r >> 2)) * ldc;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float* f = row + (long long)(n0 + 32 * j + l31) * TAPS;
            float v[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[t] = acc[t * NJ + j][r];
            if (accumulate) {                  // (all loads of the read-modify-write before the first store)
                float old[TAPS];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) old[t] = f[t];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) v[t] += old[t];
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) f[t] = v[t];
        }
    }
}

template <int TAPS, int CIB>
int wgemm_launch_t(const WGemmArgs& a, double flops, double bytes, hipStream_t s)
{
    constexpr size_t lds = (size_t)WST * (WBM * WGK + TAPS * (CIB / 32) * 32 * WGK) * sizeof(float);
    static_assert(lds <= 160 * 1024, "three stages must fit the 160 KB of LDS");
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgemm_kernel<TAPS, CIB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    TraceScope ts(K_SGEMM, s, flops, bytes);
    mcvc_launch(wgemm_kernel<TAPS, CIB>, dim3((unsigned)(a.nt * a.mt * a.nsplit)), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

int mcvc_wgemm_cib(int taps) { return taps == 1 ? 64 : 32; }

int mcvc_wgemm_launch(const WGemmArgs& a0, hipStream_t s)
{
    WGemmArgs a = a0;
    if (a.nsplit < 1) a.nsplit = 1;
    const int cib = mcvc_wgemm_cib(a.ntaps);
    if (!a.a || !a.b || !a.c || (a.M % WBM) != 0 || (a.Cin % cib) != 0 || (a.OW & 3) || a.P < a.OW || (a.P % a.OW) != 0 || a.NPIX < 4 || (a.NPIX & 3) ||
        (a.a_cs & 3) || (a.a_sn & 3) || (a.a_pitch & 3) || (a.nsplit > 1 && !a.c_slab) || (a.ntaps != 9 && a.ntaps != 3 && a.ntaps != 1))
        return MCVC_ERR_INVALID;
    if (!a.c2) { a.c2 = a.c; a.m_split = a.M; }
    if (a.m_split & 31) return MCVC_ERR_INVALID;                 // (a wave's 32 rows lie on one side of it)
    a.nstages = cdiv_i(a.NPIX, WGK);
    if (a.nsplit > a.nstages) return MCVC_ERR_INVALID;            // (the caller sums ITS slab count with dw_accum: never clamp behind its back -- ADVICE r5)
    a.nt = a.Cin / cib; a.mt = a.M / WBM;
    static const int xcd_order = mcvc_knob("MCVC_WGEMM_XCD", 1);
    a.xcd_order = xcd_order;
    a.ldc = (long long)a.Cin * a.ntaps;
    const double K = (double)a.NPIX;
    const double flops = 2.0 * a.M * a.Cin * a.ntaps * K;
    // operands once from HBM (dY; x with the taps' overlap in L2) + the gradient (read-modify-write) / the slabs
    const double bytes = 4.0 * (K * a.M + 2.25 * K * a.Cin + (double)a.M * a.Cin * a.ntaps * (a.nsplit > 1 ? a.nsplit : 2));
    if (a.ntaps == 9) return wgemm_launch_t<9, 32>(a, flops, bytes, s);
    if (a.ntaps == 3) return wgemm_launch_t<3, 32>(a, flops, bytes, s);
    return wgemm_launch_t<1, 64>(a, flops, bytes, s);
}
