// Staged-GEMM path of the discriminators' stride-2 3x3 convolutions at large batch: see sgemm.h.
#include "sgemm.h"
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ float zmasked(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }
// A 16-byte LDS read the compiler does not see as a memory access: hipcc puts `s_waitcnt vmcnt(0)` in front of a ds_read_b128 that follows
// global_load_lds instructions (it cannot tell the ring's stages apart) -- which would drain the whole LDS-DMA pipeline every stage; the
// stage's data IS complete here (explicit vmcnt wait + barrier above).  Such reads are issued as inline assembly on this LDS byte address.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }

// C = A^T B on 64 x 64 tiles, 32-deep stages, 4-stage LDS-DMA pipeline (global_load_lds, explicit vmcnt), v_mfma_f32_32x32x2_f32:
// the pipeline of gemm2_kernel<64,64,32,4> (wino_kernels.hip) with generalised operand addressing -- two A sources along K (the value
// and gate weight tensors), column-segmented B and C (image layouts [b][c][p] read / written in place), bias, and a K-split grid axis.
constexpr int BM = 64, BN = 64, GK = 32, ST = 4;
constexpr int SA = GK * BM, SB = GK * BN, STAGE = SA + SB;
constexpr int NA = SA / 4 / 256, NBI = SB / 4 / 256, ND = NA + NBI;          // DMA instructions per wave and stage (2 + 2)

__global__ void __launch_bounds__(256) sgemm_kernel(const Twin<SGemmArgs> tw)
{
    const SGemmArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int lid;
    {   // consecutive tiles (same A panel) on the same XCD: they share its L2
        const int total = (int)gridDim.x, linear = (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = linear & 7, k = linear >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n0 = (lid % a.nt) * BN; lid /= a.nt;
    const int m0 = (lid % a.mt) * BM;
    const int ks = lid / a.mt;
    const int Kc = a.K / a.nsplit;
    const int kbase = ks * Kc;
    long long aoff[NA], boff[NBI]; int adst[NA], bdst[NBI];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + i * 256;
        aoff[i] = (long long)(f / (BM / 4)) * a.lda + m0 + 4 * (f % (BM / 4));
        adst[i] = (wave * 64 + i * 256) * 4;                       // wave-uniform LDS base (float index) of this instruction
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int f = tid + i * 256;
        int n = n0 + 4 * (f % (BN / 4));
        if (n > a.N - 4) n = a.N - 4;                              // columns past N: any valid address (masked at the store)
        boff[i] = (long long)(f / (BN / 4)) * a.ldb + (long long)(n / a.bseg) * a.b_sn + (n % a.bseg);
        bdst[i] = SA + (wave * 64 + i * 256) * 4;
    }
    auto issue = [&](int stage_k, int buf) {
        float* base = smem + buf * STAGE;
        const int k0 = kbase + stage_k * GK;
        const float* ab = (k0 < a.k_split) ? a.a + (long long)k0 * a.lda : a.a2 + (long long)(k0 - a.k_split) * a.lda;
        const float* bb = a.b + (long long)k0 * a.ldb;
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(ab + aoff[i], base + adst[i]);
#pragma unroll
        for (int i = 0; i < NBI; ++i) glds16(bb + boff[i], base + bdst[i]);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nst = Kc / GK;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nst) issue(s, s);
    const int a_lane = half * BM + wm * 32 + l31;                  // A[k = 2p + half][m]
    const int b_lane = SA + half * BN + wn * 32 + l31;             // B[k = 2p + half][n]
    for (int st = 0; st < nst; ++st) {
        const int newer = (nst - 1 - st) < (ST - 2) ? (nst - 1 - st) : (ST - 2);
        if (newer >= 2) wait_vm<2 * ND>(); else if (newer == 1) wait_vm<ND>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + ST - 1 < nst) issue(st + ST - 1, (st + ST - 1) % ST);
        const float* sb = smem + (st % ST) * STAGE;
        float a0 = sb[a_lane], b0 = sb[b_lane];
#pragma unroll
        for (int p = 0; p < GK / 2; ++p) {
            const int q = (p + 1 < GK / 2) ? p + 1 : p;
            const float na0 = sb[a_lane + q * 2 * BM], nb0 = sb[b_lane + q * 2 * BN];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            a0 = na0; b0 = nb0;
        }
    }
    // (every kernarg field the stores use is read once, in front of them: left inside the 16-way unrolled, branchy loop they were re-loaded
    // with an s_waitcnt per row)
    const int n = n0 + wn * 32 + l31;
    const long long ldc = a.ldc;
    const int m_split = a.m_split, accumulate = a.accumulate;
    const float* bias = a.bias;
    float* const c1 = a.c; float* const c2 = a.c2;
    if (n < a.N) {
        const long long coff = (long long)(n / a.cseg) * a.c_sn + (n % a.cseg);
        const int mb = m0 + wm * 32 + 4 * half;
        if (ks) {
            float* slab = a.c_slab + (long long)(ks - 1) * a.c_split + (long long)mb * ldc + coff;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[r];
        } else {
            // a 32-row block lies on one side of m_split (both are multiples of 32 where two destinations are used)
            float* row0 = (mb < m_split) ? c1 + (long long)mb * ldc : c2 + (long long)(mb - m_split) * ldc;
            row0 += coff;
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[mb + (r & 3) + 8 * (r >> 2)] : 0.f;
            if (accumulate) {                    // (all 16 loads of the read-modify-write before the first store)
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = row0[(long long)((r & 3) + 8 * (r >> 2)) * ldc];
#pragma unroll
                for (int r = 0; r < 16; ++r) row0[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = old[r] + acc[r] + bv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) row0[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[r] + bv[r];
            }
        }
    }
}


// ---- implicit GEMM: sgemm_kernel with a gathered B operand, per-class A / C and a strided (scatter) C store -- see sgemm.h ----------------
// AROW (r5, the data gradient): the A operand is read from a ROW-major source -- rows = m, the 32 k of a stage contiguous -- which is what the
// tap-major FORWARD copy of the weights is to the data gradient (row (tap, ci), output channels contiguous: m = ci, k = co).  The tile is
// staged as [64 rows][8 pieces of 16 bytes] (piece p of row r at slot p ^ ((r >> 1) & 7): the lane that fills LDS slot s fetches the piece
// that belongs there) and the transposition happens in the operand read: the k slots of the 16 MFMAs of a stage are pixel... channel
// 16 * half + j (an MFMA's k index is only a label -- B is read at the same rows), so a lane needs 16 consecutive floats of its row = four
// conflict-free ds_read_b128 (wgemm_kernels.hip has the bank argument).  The per-class data-gradient copies of r4 are gone.
// ZFIX: the 1-D trunk's shifted windows over dense rows (see zpos below) -- its own instantiation, so that the discriminators' loops carry no select.
template <bool AROW, bool ZFIX>
__global__ void __launch_bounds__(256) igemm_kernel(const Twin<IGemmArgs> tw)
{
    const IGemmArgs& a = tw.v[blockIdx.z];
    const IGemmClass& cl = a.cls[gridDim.y - 1 - blockIdx.y];        // (classes are listed by ascending tap count: the longest first)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int lid;
    {   // consecutive tiles (same A panel) on the same XCD: they share its L2
        const int total = (int)gridDim.x, linear = (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = linear & 7, k = linear >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int nt = a.nt, mt = a.mt, nsplit = a.nsplit, Cb = a.Cb, OW = a.OW, P = a.P, N = a.N;
    const long long a_ks = a.a_ks, b_cs = a.b_cs, b_sn = a.b_sn;
    const int b_pitch = a.b_pitch;
    // Tile order inside a K split: groups of `mg` row tiles whose weight panels fit an XCD's L2 together; within a group the row tile is the
    // fastest index, then the column tile -- the workgroups that share a B window (the gathered activation: the large operand at large
    // batch) run side by side on one XCD and fetch it ONCE, instead of once per row tile (r5: FETCH_SIZE of this kernel at bs=32 was 4-16x
    // the activation bytes with the column tile fastest).
    const int mg = a.mgroup;
    int m_t, n_t;
    if (mg == 0) { n_t = lid % nt; lid /= nt; m_t = lid % mt; lid /= mt; }
    else {
        const int per = mt * nt;
        int r = lid % per;
        const int g = r / (mg * nt);                 // full groups first; the last group may be narrower
        const int gm = (g + 1) * mg <= mt ? mg : mt - g * mg;
        r -= g * mg * nt;
        n_t = r / gm; m_t = g * mg + r % gm;
        lid /= per;
    }
    const int n0 = n_t * BN;
    const int m0 = m_t * BM;
    const int ks = lid;
    const int Kc = cl.ntaps * Cb / nsplit;
    const int kbase = ks * Kc;
    const float* const A = cl.a;
    const float* const Bp = a.b;
    long long aoff[NA], boff[NBI]; int adst[NA], bdst[NBI];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + i * 256;
        if (AROW) {                                                // slot f of the [64 rows][8 pieces] tile: row f / 8, swizzled piece
            const int row = f >> 3, pc = (f & 7) ^ ((row >> 1) & 7);
            aoff[i] = (long long)(m0 + row) * a_ks + 4 * pc;
        } else
        aoff[i] = (long long)(f / (BM / 4)) * a_ks + m0 + 4 * (f % (BM / 4));
        adst[i] = (wave * 64 + i * 256) * 4;
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int f = tid + i * 256;
        int n = n0 + 4 * (f % (BN / 4));
        if (n > N - 4) n = N - 4;                                  // columns past N: any valid address (masked at the store)
        const int bb = n / P, rem = n - bb * P;
        const int ii = rem / OW, jj = rem - ii * OW;
        boff[i] = (long long)(f / (BN / 4)) * b_cs + (long long)bb * b_sn + (long long)ii * b_pitch + jj;
        bdst[i] = SA + (wave * 64 + i * 256) * 4;
    }
    auto issue = [&](int stage_k, int buf) {
        float* base = smem + buf * STAGE;
        const int k0 = kbase + stage_k * GK;
        const int t = k0 / Cb, c0 = k0 - t * Cb;                  // (a stage lies inside one tap: Cb % 32 == 0)
        const float* ab = AROW ? A + cl.aoff[t] + c0 : A + cl.aoff[t] + (long long)c0 * a_ks;
        const float* bb = Bp + cl.boff[t] + (long long)c0 * b_cs;
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(ab + aoff[i], base + adst[i]);
#pragma unroll
        for (int i = 0; i < NBI; ++i) glds16(bb + boff[i], base + bdst[i]);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nst = Kc / GK;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nst) issue(s, s);
    const int a_lane = half * BM + wm * 32 + l31;                  // A[k = 2p + half][m]
    const int b_lane = SA + half * BN + wn * 32 + l31;             // B[k = 2p + half][n]
    // 1-D convolutions over DENSE rows of zw columns (the trunk, sgemm.h): a tap's window shifted by -1 / +1 column reads the neighbouring row's
    // last / first element at a row's first / last column -- this lane's B values of such a tap are replaced by the zero the padding holds
    int zpos = 0;                                                  // 1: first column of a row, 2: last, 0: neither (or no fix)
    if (ZFIX) { const int w = (n0 + wn * 32 + l31) % a.zw; zpos = (w == 0) ? 1 : (w == a.zw - 1 ? 2 : 0); }
    for (int st = 0; st < nst; ++st) {
        const int newer = (nst - 1 - st) < (ST - 2) ? (nst - 1 - st) : (ST - 2);
        if (newer >= 2) wait_vm<2 * ND>(); else if (newer == 1) wait_vm<ND>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + ST - 1 < nst) issue(st + ST - 1, (st + ST - 1) % ST);
        const float* sb = smem + (st % ST) * STAGE;
        // (ZFIX) this stage's tap: 1 = window shifted by -1 column, 2 = by +1; a lane at that end of a row takes 0 instead of the
        // neighbouring row's element -- a bit mask on the LOADED value (v_and_b32), not a branch and not a scale: the loads stay
        // unconditional, and the float beyond the tensor's first / last row may be anything (NaN, Inf: 0 * NaN would be NaN -- ADVICE r5)
        unsigned zmask = 0xffffffffu;
        if (ZFIX) { const int zt = cl.zs[(kbase + st * GK) / Cb]; zmask = (zpos != 0 && zt == zpos) ? 0u : 0xffffffffu; }
        if (AROW) {
            // operand reads one group of four MFMAs ahead (one ds_read_b128 of A + four ds_read_b32 of B per group), order pinned
            const float* ar = sb + (wm * 32 + l31) * GK;
            const int sw = (l31 >> 1) & 7;
            const float* br = sb + SA + (16 * half) * BN + wn * 32 + l31;
            // (all operand reads of this branch are inline assembly -- see lds_read16 -- with their own waits: a group's reads are requested
            //  in front of the previous group's four MFMAs and awaited behind them)
            const unsigned a_addr = lds_addr(ar), b_addr = lds_addr(br);
            f32x4 av, an; float b0, b1, b2, b3, n0_, n1_, n2_, n3_;
            asm volatile("ds_read_b128 %0, %1" : "=&v"(av) : "v"(a_addr + ((((4 * half) ^ sw)) << 4)));
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(b_addr));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                if (q < 3) {
                    asm volatile("ds_read_b128 %0, %1" : "=&v"(an) : "v"(a_addr + ((((4 * half + q + 1) ^ sw)) << 4)));
                    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768"
                                 : "=&v"(n0_), "=&v"(n1_), "=&v"(n2_), "=&v"(n3_) : "v"(b_addr + (q + 1) * 1024));
                }
                __builtin_amdgcn_sched_barrier(0);        // (the four MFMAs below stay BEHIND the next group's requests)
                if (ZFIX) { b0 = zmasked(b0, zmask); b1 = zmasked(b1, zmask); b2 = zmasked(b2, zmask); b3 = zmasked(b3, zmask); }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);        // (... and the next group's wait stays behind them)
                if (q < 3) { av = an; b0 = n0_; b1 = n1_; b2 = n2_; b3 = n3_; }
            }
            continue;
        }
        float a0 = sb[a_lane], b0 = sb[b_lane];
#pragma unroll
        for (int p = 0; p < GK / 2; ++p) {
            const int q = (p + 1 < GK / 2) ? p + 1 : p;
            const float na0 = sb[a_lane + q * 2 * BM], nb0 = sb[b_lane + q * 2 * BN];
            if (ZFIX) b0 = zmasked(b0, zmask);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            a0 = na0; b0 = nb0;
        }
    }
    const int n = n0 + wn * 32 + l31;
    const long long ldc = a.ldc;
    const int accumulate = a.accumulate;
    const float* bias = a.bias;
    if (n < N) {
        const int bb = n / P, rem = n - bb * P;
        const int ii = rem / OW, jj = rem - ii * OW;
        const int mb = m0 + wm * 32 + 4 * half;
        const long long coff = cl.coff + (long long)bb * a.c_sn + (long long)ii * a.c_sh + (long long)jj * a.c_sw + (long long)mb * ldc;
        if (ks) {
            float* slab = a.c_slab + (long long)(ks - 1) * a.c_split + coff;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[r];
        } else {
            float* row0 = a.c + coff;
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[mb + (r & 3) + 8 * (r >> 2)] : 0.f;
            if (accumulate) {                    // (all 16 loads of the read-modify-write before the first store)
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = row0[(long long)((r & 3) + 8 * (r >> 2)) * ldc];
#pragma unroll
                for (int r = 0; r < 16; ++r) row0[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = old[r] + acc[r] + bv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) row0[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[r] + bv[r];
            }
        }
    }
}

// ---- the weight gradients' K-split slabs ----------------------------------------------------------------------------------------------
struct DwAccumKArgs { const float* slabs; int nslab; long long slab_stride; float* g0; float* g1; int Cout; int rows; int K9; };
__global__ void __launch_bounds__(256) dw_accum_kernel(const Twin<DwAccumKArgs> tw)
{
    const DwAccumKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ slabs = ka_.slabs;
    int nslab = ka_.nslab;
    long long slab_stride = ka_.slab_stride;
    float* __restrict__ g0 = ka_.g0;
    float* __restrict__ g1 = ka_.g1;
    int Cout = ka_.Cout;
    int rows = ka_.rows;
    int K9 = ka_.K9;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long long)rows * K9) return;
    float4 s = *reinterpret_cast<const float4*>(slabs + i);
    for (int k = 1; k < nslab; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (long long)k * slab_stride + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long long split = (long long)Cout * K9;
    float4* d = reinterpret_cast<float4*>(i < split ? g0 + i : g1 + (i - split));
    float4 o = *d;
    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
    *d = o;
}


// ---- layout converters for the op-level entries (the networks' producers write these layouts directly) -------------------------------------
struct XsConvKArgs { const float* src; float* dst; int C, H, W, pw; long long plane; int mode; };       // mode 0: xs, 1: padded dY
__global__ void __launch_bounds__(256) layout_conv_kernel(const Twin<XsConvKArgs> tw)
{
    const XsConvKArgs a = tw.v[blockIdx.z];
    const long long nc = blockIdx.y;                                   // (sample, channel)
    if (a.mode == 0) {
        const long long tot = 4 * a.plane;
        float* d = a.dst + nc * tot;
        const float* sp = a.src + nc * a.H * a.W;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) {
            const int pq = (int)(i / a.plane); const int rem = (int)(i - (long long)pq * a.plane);
            const int r = rem / a.pw, c = rem - r * a.pw;
            float v = 0.f;
            if (r >= 1 && c >= 4) { const int h = 2 * (r - 1) + (pq >> 1), w = 2 * (c - 4) + (pq & 1); if (h < a.H && w < a.W) v = sp[(long long)h * a.W + w]; }
            d[i] = v;
        }
    } else {
        float* d = a.dst + nc * a.plane;
        const float* sp = a.src + nc * a.H * a.W;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.plane; i += (long long)gridDim.x * 256) {
            const int r = (int)(i / a.pw), c = (int)(i - (long long)r * a.pw);
            d[i] = (r < a.H && c < a.W) ? sp[(long long)r * a.W + c] : 0.f;
        }
    }
}

}  // namespace

int mcvc_xs_from_dense_launch(const float* x, float* xs, int NB, int C, int H, int W, hipStream_t s)
{
    if ((H & 1) || (W & 1)) return MCVC_ERR_INVALID;
    const long long plane = mcvc_xs_plane(H, W);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * NB * C * ((double)H * W + 4.0 * plane));
    mcvc_launch(layout_conv_kernel, dim3((unsigned)((4 * plane + 255) / 256), (unsigned)(NB * C)), dim3(256), 0, s, XsConvKArgs{x, xs, C, H, W, mcvc_xs_pw(W), plane, 0});
    return (int)hipGetLastError();
}

int mcvc_dyp_from_dense_launch(const float* dy, float* dyp, int NB, int C, int OH, int OW, hipStream_t s)
{
    const long long plane = mcvc_dyp_plane(OH, OW);
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * NB * C * ((double)OH * OW + plane));
    mcvc_launch(layout_conv_kernel, dim3((unsigned)((plane + 255) / 256), (unsigned)(NB * C)), dim3(256), 0, s, XsConvKArgs{dy, dyp, C, OH, OW, mcvc_dyp_pitch(OW), plane, 1});
    return (int)hipGetLastError();
}

int mcvc_sgemm_launch(const SGemmArgs& a0, hipStream_t s)
{
    SGemmArgs a = a0;
    if (a.nsplit < 1) a.nsplit = 1;
    if ((a.M % BM) != 0 || (a.K % (GK * a.nsplit)) != 0 || (a.N & 3) || a.N < 4 || (a.lda & 3) || (a.ldb & 3) || (a.bseg & 3) || (a.b_sn & 3) ||
        a.bseg < 4 || a.cseg < 1)
        return MCVC_ERR_INVALID;
    if (!a.a2) { a.a2 = a.a; a.k_split = a.K; }
    if (!a.c2) { a.c2 = a.c; a.m_split = a.M; }
    if (a.nsplit > 1 && !a.c_slab) return MCVC_ERR_INVALID;
    if (a.m_split & 31) return MCVC_ERR_INVALID;                 // (the store picks the destination per 32-row block)
    a.nt = cdiv_i(a.N, BN); a.mt = a.M / BM;
    constexpr size_t lds = (size_t)ST * STAGE * sizeof(float);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sgemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    TraceScope ts(K_SGEMM, s, 2.0 * a.M * a.N * a.K, 4.0 * ((double)a.K * a.M + (double)a.K * a.N + (double)a.M * a.N * a.nsplit));
    mcvc_launch(sgemm_kernel, dim3((unsigned)(a.nt * a.mt * a.nsplit)), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

int mcvc_igemm_launch(const IGemmArgs& a0, hipStream_t s)
{
    IGemmArgs a = a0;
    if (a.nsplit < 1) a.nsplit = 1;
    if (a.ncls < 1 || a.ncls > 4 || (a.M % BM) != 0 || (a.N & 3) || a.N < 4 || (a.a_ks & 3) || (a.Cb % GK) != 0 || (a.OW & 3) || a.P < a.OW || (a.P % a.OW) != 0 ||
        (a.nsplit > 1 && !a.c_slab))
        return MCVC_ERR_INVALID;
    double flops = 0.0, bytes = 0.0;
    for (int c = 0; c < a.ncls; ++c) {
        const IGemmClass& k = a.cls[c];
        if (!k.a || k.ntaps < 1 || k.ntaps > 9 || ((long long)k.ntaps * a.Cb) % (GK * a.nsplit) != 0) return MCVC_ERR_INVALID;
        const double K = (double)k.ntaps * a.Cb;
        flops += 2.0 * a.M * a.N * K;
        bytes += 4.0 * (K * a.M + (double)a.M * a.N * a.nsplit);
    }
    bytes += 4.0 * (double)a.Cb * a.N * 2.25;            // (the gathered activation: read once from HBM, the taps' overlap hits in L2)
    a.nt = cdiv_i(a.N, BN); a.mt = a.M / BM;
    {   // Row tiles per group of the tile order (igemm_kernel).  Two kinds of reuse decide what an XCD fetches through its L2:
        //  * over TIME -- a group's A panels (the longest class's K x 64 floats each) stay in L2 while the group's column tiles pass by: needs
        //    the panels within ~2 MB of the 4 MB (r5's rule; right for the short grids: a whole XCD's share is resident at once and A dominates);
        //  * SIMULTANEOUS -- the ~64 workgroups an XCD holds (2 per CU) sweep k side by side, so a stage's A chunk is fetched once for all the
        //    column tiles of its row among them and its B chunk once for all the rows of its column: gm x (64 / gm) tiles fetch
        //    (gm + 64 / gm) chunks per stage -- 16 at gm = 8 against 65 at gm = 1 -- and need no capacity at all.  r5's rule gives gm = 1 exactly
        //    where it matters (downSample3: 1.2 MB panels) and the gathered activation was fetched once per ROW tile: FETCH_SIZE per launch at
        //    32 samples 626 -> 252 MB (forward), 338 -> 262 MB (data gradient) with 8 rows per group; at 8 samples (an XCD's share is 20-80
        //    tiles, all resident at once) the data gradient went 42 -> 82 MB, so the grid must hold at least one full resident round per XCD
        //    (profiles/r06b_pmc_order.log; the step time does not move either way: r06b_ab_tile_order.log).
        int kmax = 0;
        for (int c = 0; c < a.ncls; ++c) kmax = a.cls[c].ntaps * a.Cb > kmax ? a.cls[c].ntaps * a.Cb : kmax;
        static const int l2kb = mcvc_knob("MCVC_IGEMM_GROUP_KB", 2048);          // (0: the column tile fastest, r4's order)
        static const int simul = mcvc_knob("MCVC_IGEMM_GROUP_ROWS", 8);          // (0: r5's rule alone)
        const long long panel = (long long)kmax * BM * 4;
        long long mg = l2kb > 0 ? (long long)l2kb * 1024 / (panel > 0 ? panel : 1) : 0;
        if (l2kb > 0 && mg < 1) mg = 1;
        if (l2kb > 0 && simul > mg && (long long)a.nt * a.mt * a.nsplit >= 8 * 64) mg = simul;
        if (mg > a.mt) mg = a.mt;
        a.mgroup = (int)mg;
    }
    constexpr size_t lds = (size_t)ST * STAGE * sizeof(float);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    TraceScope ts(K_SGEMM, s, flops, bytes);
    const dim3 grid((unsigned)(a.nt * a.mt * a.nsplit), (unsigned)a.ncls);
    if (a.zw > 0) {
        if (a.arow) mcvc_launch(igemm_kernel<true, true>, grid, dim3(256), lds, s, a);
        else mcvc_launch(igemm_kernel<false, true>, grid, dim3(256), lds, s, a);
    } else {
        if (a.arow) mcvc_launch(igemm_kernel<true, false>, grid, dim3(256), lds, s, a);
        else mcvc_launch(igemm_kernel<false, false>, grid, dim3(256), lds, s, a);
    }
    return (int)hipGetLastError();
}

int mcvc_dw_accum_launch(const float* slabs, int nslab, long long slab_stride, float* g0, float* g1, int Cout, int rows, int K9, hipStream_t s)
{
    const long long n = (long long)rows * K9;
    if ((n & 3) || (((long long)Cout * K9) & 3) || (rows > Cout && !g1)) return MCVC_ERR_INVALID;
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * n * (nslab + 2.0));
    mcvc_launch(dw_accum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, DwAccumKArgs{slabs, nslab, slab_stride, g0, g1, Cout, rows, K9});
    return (int)hipGetLastError();
}
