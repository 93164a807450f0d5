// bf16 inference kernels of the Generator for gfx950 (MI355X): NHWC bf16 activations, v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, fp32 InstanceNorm statistics.  Replaces, for `generator(real, ones)` of the reference's inference driver
// (mask_cyclegan_vc/test.py:85-119 -> model.py:239-280), the fp32 path when the caller asks for bf16 (BASELINE configs[4]).
//
// Convolution = implicit GEMM, im2col-free:  M = output channel, N = output pixel, K = (kh, ci-chunk, kw, ci).
//   * activations are channel-innermost, so the 8 consecutive k of one MFMA operand lane are 8 consecutive input channels of
//     ONE pixel: a 16-byte LDS read; every tap (kh, kw) of the filter reads a SHIFTED window of the same staged input patch.
//   * LDS rows are 64 bytes (32 channels) per pixel / per (weight row, tap); the four 16-byte chunks of a row are XOR-swizzled
//     with bits 2-3 of the row index, which makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte bank slots
//     (stride-1 pixels and odd KW; stride-2 pixel reads are 2-way).
//   * weights stream per (kh, ci-chunk) stage by LDS-DMA (global_load_lds, 16 B per lane) into a double buffer: the copy of stage
//     s+1 is in flight while the MFMAs of stage s run; one s_waitcnt vmcnt(0) + __syncthreads() per stage.  A stage's 2*KW
//     (tap, 16-channel) steps are software-pipelined: the operands of step s+1 are read while step s multiplies.
#include "mcvc_common.h"
#include "bf16.h"
#include "trace.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

// fp32 -> bf16, round to nearest even, NaN stays NaN: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per instruction;
// the integer emulation it replaces cost ~12 instructions and an exec-mask branch per value -- a third of the conv epilogue)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float a, float b)
{
    const bf16x2_ r = __builtin_convertvector(f32x2_{a, b}, bf16x2_);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
// (bf16 path: the result is rounded to 8 mantissa bits -- hardware exp2 / reciprocal (1 ulp) instead of the correctly rounded expf and an IEEE
//  division: the GLU apply pass of downSample1 was bound by these ~60 instructions per element, norm family 0.575 -> 0.53 ms per 16 x 512-frame forward)
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f)
{
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float* f)
{
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ======================================================================================================================
// convolution
// ======================================================================================================================
constexpr int kConvThreads = 256;
constexpr int kMaxWP = 10;          // 16-byte weight pieces per thread and stage (BM * KW * 4 <= 2560)
// 16-byte input-patch pieces per thread that are prefetched through registers one channel chunk ahead: template parameter PPT
// (4 for stride-1 layers, 12 for the stride-2 layers whose patch is ~4x larger)

// KWT > 0: the kernel width is a compile-time constant (taps fully unrolled: the compiler hoists the next operands' LDS reads above
// the current MFMAs); KWT == 0: run-time width.
// (Weights through registers instead of LDS-DMA -- global -> registers -> ds_write_b128 -- was built in r4 and measured slower: 938 -> 1004 us on
// upSample2, 20-47 spilled registers and the stores exposed in front of the stage barrier; removed)
template <int WM, int WN, int MT, int NT, int KWT, int PPT>
__global__ void __launch_bounds__(kConvThreads) bf16_conv_kernel(const Bf16ConvArgs a)
{
    constexpr int kMaxPP = PPT > 0 ? PPT : 1;
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int KW = KWT > 0 ? KWT : a.KW;

    // ---- block -> (output-channel tile, pixel tile).  Workgroup b runs on XCD b % 8 (observed placement, speed only): when the
    // number of channel tiles divides 8, the workgroups of one XCD all use ONE weight slice, which then stays in that XCD's L2.
    const int n_co = a.Cout_pad / BM;
    const int n_px = a.N * a.tiles_h * a.tiles_w;
    int co_tile, px_tile;
    {
        const int b = blockIdx.x;
        if (n_co <= 8 && (8 % n_co) == 0 && (n_px % (8 / n_co)) == 0) {
            const int xcd = b & 7, j = b >> 3, per = 8 / n_co;
            co_tile = xcd % n_co;
            px_tile = j * per + xcd / n_co;
        } else {
            co_tile = b % n_co;
            px_tile = b / n_co;
        }
    }
    const int co0 = co_tile * BM;
    const int TW = 1 << a.tw_log2;
    const int n_img = px_tile / (a.tiles_h * a.tiles_w);
    const int trem = px_tile - n_img * (a.tiles_h * a.tiles_w);
    const int oh0 = (trem / a.tiles_w) * a.TH, ow0 = (trem % a.tiles_w) * TW;
    const int ih0 = oh0 * a.stride - a.pad_h, iw0 = ow0 * a.stride - a.pad_w;

    const int patch_px = a.PH * a.PW;
    const int patch_pieces = patch_px * 4;
    unsigned char* Xs = smem;                                             // [patch_px][64 B]
    const int wbuf_bytes = BM * KW * 64;
    unsigned char* Ws = smem + a.patch_bytes;                             // wbufs x [BM][KW][64 B]

    const int ncc = a.Cin >> 5;
    const int nstage = ncc * a.KH;                                        // stage = (cc, kh): KW taps x 32 input channels
    const int w_pieces = BM * KW * 4;                                     // 16-byte pieces per weight stage
    // Weights go HBM / L2 -> LDS by direct DMA (global_load_lds, 16 B per lane, no registers) into a double buffer: the copy of stage s+1
    // is in flight during the MFMAs of stage s.  (Round 2 staged them through a register array that the compiler had placed in SCRATCH
    // memory -- global load -> wait -> scratch store -> scratch load -> LDS store per stage, the load latency fully exposed: an in-kernel
    // clock showed 1.3 us of store + 4.8 us "MFMA" phase per stage for 1.1 us of MFMA work.)
    constexpr int NWP = KWT > 0 ? (BM * KWT * 4 + kConvThreads - 1) / kConvThreads : kMaxWP;
    // piece -> (weight row, tap, swizzled chunk) is stage-invariant: element offsets computed once
    constexpr bool W_EXACT = KWT > 0 && (BM * KWT * 4) % kConvThreads == 0;      // every lane of every piece is a real piece: no exec masking
    int woff[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int q = tid + i * kConvThreads;
        const int row = q / (KW * 4), rem = q - row * (KW * 4);
        const int tap = rem >> 2, pos = rem & 3;
        const int lc = pos ^ ((row >> 2) & 3);
        woff[i] = (q < w_pieces) ? ((co0 + row) * a.KH * ncc * KW + tap) * 32 + lc * 8 : -1;
    }
    auto w_src = [&](int stage) { const int cc = stage / a.KH, kh = stage - cc * a.KH; return a.w + (long long)(kh * ncc + cc) * KW * 32; };
    auto issue_piece = [&](const bf16_t* base, unsigned char* dst, int i) __attribute__((always_inline)) {
        if (W_EXACT || woff[i] >= 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + woff[i]),
                                             (__attribute__((address_space(3))) void*)(dst + (size_t)(wave * 64 + i * kConvThreads) * 16), 16, 0, 0);
    };
    auto issue_w = [&](int stage, int buf) {
        const bf16_t* base = w_src(stage);
        unsigned char* dst = Ws + buf * wbuf_bytes;
#pragma unroll
        for (int i = 0; i < NWP; ++i) issue_piece(base, dst, i);
    };
    // ---- input patch: small patches (<= kMaxPP pieces per thread) are prefetched through registers one chunk ahead; large ones
    //      (stride-2 layers) are staged synchronously, four loads in flight per thread
    // stride 2: patch columns are stored de-interleaved (even columns, then odd columns) so that the 16 lanes of an operand read --
    // output columns c, c+1, ... = input columns 2c+kw, 2c+2+kw, ... -- walk CONSECUTIVE LDS pixels (conflict-free with the swizzle)
    const bool s2 = a.stride == 2;
    const int PWe = (a.PW + 1) >> 1;
    auto patch_col = [&](int j) { return s2 ? (j < PWe ? 2 * j : 2 * (j - PWe) + 1) : j; };
    // PPT > 0: the host guarantees patch_pieces <= PPT * 256 and the patch is ALWAYS prefetched; PPT == 0: always staged synchronously.  (One
    // kernel with both paths behind a run-time test made the compiler treat the synchronous loop's load registers as possibly pending at the
    // first operand read of every stage: an s_waitcnt vmcnt(0) there, i.e. in front of the MFMAs and behind the weight DMA just issued.)
    constexpr bool p_pref = PPT > 0;
    int poff[kMaxPP];                         // element offset of the piece inside the image (without the chunk offset; an image is < 2^31 elements: host check)
    unsigned pvalid = 0;                      // bit i: piece i is inside the image (else it is loaded from offset 0 and replaced by zeros at the LDS store:
                                              // loads and stores without exec-mask branches)
#pragma unroll
    for (int i = 0; i < kMaxPP; ++i) {
        const int q = tid + i * kConvThreads;
        const int pp = q >> 2, pos = q & 3;
        const int pr = pp / a.PW, pc = pp - pr * a.PW;
        const int ih = ih0 + pr, iw = iw0 + patch_col(pc);
        const bool ok = p_pref && q < patch_pieces && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        poff[i] = ok ? (int)((long long)ih * a.x_sh + (long long)iw * a.x_sw + (pos ^ ((pp >> 2) & 3)) * 8) : 0;
        pvalid |= ok ? (1u << i) : 0u;
    }
    const bf16_t* ximg = a.x + (long long)n_img * a.x_sn;
    uint4 preg[kMaxPP];
    auto load_patch = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kMaxPP; ++i)
            preg[i] = *reinterpret_cast<const uint4*>(ximg + cc * 32 + poff[i]);
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kMaxPP; ++i) {
            const int q = tid + i * kConvThreads;                  // q * 16 < kMaxPP * 4096 <= patch_bytes (launcher)
            const bool ok = (pvalid >> i) & 1u;
            uint4 v = preg[i];
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<uint4*>(Xs + (size_t)q * 16) = v;
        }
    };
    auto stage_patch_sync = [&](int cc) {
        const bf16_t* xb = ximg + cc * 32;
        for (int q0 = tid; q0 < patch_pieces; q0 += 4 * kConvThreads) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * kConvThreads;
                const int pp = q >> 2, pos = q & 3;
                const int pr = pp / a.PW, pc = pp - pr * a.PW;
                const int ih = ih0 + pr, iw = iw0 + patch_col(pc);
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (q < patch_pieces && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                    v[u] = *reinterpret_cast<const uint4*>(xb + (long long)ih * a.x_sh + (long long)iw * a.x_sw + (pos ^ ((pp >> 2) & 3)) * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * kConvThreads;
                if (q < patch_pieces) *reinterpret_cast<uint4*>(Xs + (size_t)q * 16) = v[u];
            }
        }
    };

    // ---- per-lane operand coordinates
    int rowA[MT], gA[MT], ppB[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { rowA[mt] = (wm * MT + mt) * 32 + l31; gA[mt] = (rowA[mt] >> 2) & 3; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (wn * NT + nt) * 32 + l31;
        const int pr = n >> a.tw_log2, pc = n & (TW - 1);
        ppB[nt] = pr * a.stride * a.PW + (s2 ? pc : pc * a.stride);          // (stride 2: de-interleaved columns -> lane stride 1)
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    issue_w(0, 0);
    if constexpr (p_pref) load_patch(0);
    for (int stage = 0; stage < nstage; ++stage) {
        const int cc = stage / a.KH, kh = stage - cc * a.KH;
        const int buf = stage & 1;
        // everything requested during the previous stage (this stage's weights, the next chunk's patch registers) has had that stage's
        // MFMAs to land; the barrier also says every wave is done reading the buffer the next copy overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kh == 0) {
            if constexpr (p_pref) store_patch(); else stage_patch_sync(cc);
            __syncthreads();
        }
        // KWT > 0: the next stage's weight DMA and the next chunk's patch loads are issued from inside the step loop, behind the first MFMAs
        // (issued here they sat in front of the stage's first operand reads: ten DMA instructions with their address arithmetic per stage while
        // the matrix pipe idled)
        if constexpr (KWT == 0) {
            if (stage + 1 < nstage) issue_w(stage + 1, buf ^ 1);          // in flight during the MFMAs below
            if constexpr (p_pref) { if (kh == a.KH - 1 && cc + 1 < ncc) load_patch(cc + 1); }
        }
        const unsigned char* Wb = Ws + buf * wbuf_bytes;
        const int pk = kh * a.PW;
        // one step = (tap, 16-channel half) = MT + NT operand reads (ds_read_b128) and MT * NT MFMAs
        // weight operand address = (per-lane base of (mt, 16-channel half)) + tap * 64: the tap offset is an instruction immediate once the
        // steps are unrolled, so 2 * MT base registers serve all 2 * KW steps (the compiler otherwise keeps one per (mt, tap, half): 40)
        const unsigned char* abase[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) abase[mt][h2] = Wb + (size_t)(((rowA[mt] * KW) << 2) + ((h2 * 2 + half) ^ gA[mt])) * 16;
        auto load_step = [&](int step, bf16x8 (&av)[MT], bf16x8 (&bv)[NT]) {
            const int tap = step >> 1, lc = (step & 1) * 2 + half;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                av[mt] = *reinterpret_cast<const bf16x8*>(abase[mt][step & 1] + tap * 64);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int pp = ppB[nt] + pk + (s2 ? (tap >> 1) + (tap & 1) * PWe : tap);
                bv[nt] = *reinterpret_cast<const bf16x8*>(Xs + (size_t)((pp << 2) + (lc ^ ((pp >> 2) & 3))) * 16);
            }
        };
        auto mfma_step = [&](const bf16x8 (&av)[MT], const bf16x8 (&bv)[NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        };
        if constexpr (KWT > 0) {
            // Software pipeline over the 2 * KW steps of the stage: the operands of step s+1 are requested while step s multiplies.  The reads
            // are inline assembly: hipcc cannot tell that the LDS-DMA of the NEXT stage's weights (global_load_lds above) does not write what a
            // ds_read_b128 here reads, and put `s_waitcnt vmcnt(0)` in front of the stage's first operand read -- every stage then waited for the
            // next stage's 40 KB weight copy (and the next chunk's patch loads) before its first MFMA, the copy never overlapped the matrix work
            // (SQ: MFMA busy 0.31 - 0.53 on the large layers).  With the reads opaque the compiler keeps its hands off the counters: one explicit
            // lgkmcnt(0) per step, placed before the next step's requests are issued, and sched_barrier(0) pins requests / MFMAs / wait in order.
            const unsigned xs_addr = lds_addr(Xs), wb_addr = lds_addr(Wb);
            unsigned aaddr[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) aaddr[mt][h2] = wb_addr + (unsigned)((((rowA[mt] * KW) << 2) + ((h2 * 2 + half) ^ gA[mt])) << 4);
            // pixel operand of (nt, tap): patch pixel pp, 16-byte slot (h2 * 2 + half) ^ ((pp >> 2) & 3); the two 16-channel halves differ in bit 5
            auto b_addr = [&](int tap, unsigned (&ad)[NT][2]) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int pp = ppB[nt] + pk + (s2 ? (tap >> 1) + (tap & 1) * PWe : tap);
                    const unsigned slot = (unsigned)((half ^ ((pp >> 2) & 3)) << 4);
                    const unsigned row = xs_addr + ((unsigned)pp << 6);
                    ad[nt][0] = row + slot; ad[nt][1] = row + (slot ^ 32u);
                }
            };
            u32x4 av[2][MT], bv[2][NT];
            unsigned bad[2][NT][2];                                  // [tap & 1]
            auto request = [&](int step, u32x4 (&a_)[MT], u32x4 (&b_)[NT]) __attribute__((always_inline)) {
                const int tap = step >> 1, h2 = step & 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(a_[mt]) : "v"(aaddr[mt][h2]), "i"(tap * 64));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    asm volatile("ds_read_b128 %0, %1" : "=&v"(b_[nt]) : "v"(bad[tap & 1][nt][h2]));
            };
            auto landed = [&](u32x4 (&a_)[MT], u32x4 (&b_)[NT]) __attribute__((always_inline)) {
                if constexpr (MT == 4 && NT == 4)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(b_[0]), "+v"(b_[1]), "+v"(b_[2]), "+v"(b_[3]));
                else if constexpr (MT == 2 && NT == 4)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(b_[0]), "+v"(b_[1]), "+v"(b_[2]), "+v"(b_[3]));
                else if constexpr (MT == 2 && NT == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(b_[0]), "+v"(b_[1]));
                else if constexpr (MT == 2 && NT == 1)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(b_[0]));
                else {
                    static_assert(MT == 1 && NT == 1, "operand-wait variant missing for this tile");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(b_[0]));
                }
            };
            // The next stage's weights: NWP DMA instructions per wave, spread over the stage's first 2 * KWT - 2 steps, each in the middle of
            // a step's MFMAs.  One LDS-DMA instruction occupies its wave's issue slot for 60 - 185 cycles (MI355X_MICROARCH.md), far more than
            // the 32-cycle MFMA beside it hides, and four waves issuing ten each in one burst queue behind each other in the texture path:
            // with all ten at the head of the stage the copy cost 21 % of the layer (832 us with, 655 us without, upSample2 at 16 x 512 frames).
            // Unconditional (a branch would end the scheduling region): the block's final stage re-requests its own weights into the idle buffer.
            const bf16_t* wnext = w_src(stage + 1 < nstage ? stage + 1 : stage);
            unsigned char* dnext = Ws + (buf ^ 1) * wbuf_bytes;
            constexpr int DSTEPS = 2 * KWT > 2 ? 2 * KWT - 2 : 1;
            b_addr(0, bad[0]);
            request(0, av[0], bv[0]);
            // One step = MT * NT MFMAs.  The NEXT step's MT + NT operand reads go one per MFMA gap at the head of the step (eight reads in a
            // burst in front of the MFMAs cost +14 % in tools/ubench_issue_cost.hip, one per gap costs nothing), then the next tap's pixel
            // addresses (VALU), then this step's share of the weight DMA; a sched_barrier after every MFMA keeps that order.
            auto request_one = [&](int step, int r, u32x4 (&a_)[MT], u32x4 (&b_)[NT]) __attribute__((always_inline)) {
                const int tap = step >> 1, h2 = step & 1;
                if (r < MT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(a_[r]) : "v"(aaddr[r][h2]), "i"(tap * 64));
                else asm volatile("ds_read_b128 %0, %1" : "=&v"(b_[r - MT]) : "v"(bad[tap & 1][r - MT][h2]));
            };
            auto b_addr_one = [&](int tap, int nt) __attribute__((always_inline)) {
                const int pp = ppB[nt] + pk + (s2 ? (tap >> 1) + (tap & 1) * PWe : tap);
                const unsigned slot = (unsigned)((half ^ ((pp >> 2) & 3)) << 4);
                const unsigned row = xs_addr + ((unsigned)pp << 6);
                bad[tap & 1][nt][0] = row + slot; bad[tap & 1][nt][1] = row + (slot ^ 32u);
            };
            constexpr int MN = MT * NT, NR = MT + NT;
#pragma unroll
            for (int step = 0; step < 2 * KWT; ++step) {
                landed(av[step & 1], bv[step & 1]);
                const bool more = step + 1 < 2 * KWT;
#pragma unroll
                for (int m = 0; m < MN; ++m) {
                    if (more) {
#pragma unroll
                        for (int r = 0; r < NR; ++r)
                            if ((r < MN ? r : MN - 1) == m) request_one(step + 1, r, av[(step + 1) & 1], bv[(step + 1) & 1]);
                    }
                    acc[m / NT][m % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[step & 1][m / NT]),
                                                                                  __builtin_bit_cast(bf16x8, bv[step & 1][m % NT]), acc[m / NT][m % NT], 0, 0, 0);
                    if ((step & 1) == 0 && step + 2 < 2 * KWT) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            if ((NR + nt < MN ? NR + nt : MN - 1) == m) b_addr_one((step >> 1) + 1, nt);
                    }
#pragma unroll
                    for (int i = 0; i < NWP; ++i)
                        if (i * DSTEPS / NWP == step) {
                            int k = 0;                            // index of piece i among this step's pieces
#pragma unroll
                            for (int j = 0; j < i; ++j) k += (j * DSTEPS / NWP == step) ? 1 : 0;
                            const int at = NR + NT + 1 + 2 * k;
                            if ((at < MN ? at : MN - 1) == m) issue_piece(wnext, dnext, i);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (p_pref) {
                    if (step == (2 * KWT > 4 ? 3 : 2 * KWT - 1)) { if (kh == a.KH - 1 && cc + 1 < ncc) load_patch(cc + 1); }
                }
            }
        } else {
            for (int step = 0; step < 2 * KW; ++step) {
                bf16x8 av[MT], bv[NT];
                load_step(step, av, bv);
                mfma_step(av, bv);
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the last stage's redundant weight request)
    // the tile's BM bias values go through LDS once (per-store global loads, each behind its own `bias != nullptr` branch and vmcnt(0), were
    // a latency chain of 64 round trips per workgroup)
    __syncthreads();
    float* sbias = reinterpret_cast<float*>(smem);
    if (tid < BM) sbias[tid] = (a.bias && co0 + tid < a.Cout_pad) ? a.bias[co0 + tid] : 0.f;
    __syncthreads();
    // ---- epilogue: bias (+ fused GLU), bf16 NHWC store.  Accumulator register r of a lane is output row
    //      (r & 3) + 8 * (r >> 2) + 4 * half of the 32-row tile, column l31: four consecutive channels per r >> 2 -> 8-byte stores.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (wn * NT + nt) * 32 + l31;
        const int oh = oh0 + (n >> a.tw_log2), ow = ow0 + (n & (TW - 1));
        if (oh >= a.OH || ow >= a.OW) continue;
        bf16_t* yp = a.y + (long long)n_img * a.y_sn + (long long)oh * a.y_sh + (long long)ow * a.y_sw;
        if (a.glu) {
            if constexpr (MT == 2) {
                const int cbase = (co0 >> 1) + wm * 32;             // 64-row block = [32 value | 32 gate] rows of channels cbase..+31
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = cbase + 8 * q + 4 * half;
                    if (c < a.Cout) {
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[0][nt][4 * q + j] + sbias[wm * 64 + 8 * q + 4 * half + j];
                            const float g = acc[1][nt][4 * q + j] + sbias[wm * 64 + 32 + 8 * q + 4 * half + j];
                            o[j] = v * sigmoidf_(g);
                        }
                        uint2 pk2;
                        pk2.x = pack2bf(o[0], o[1]);
                        pk2.y = pack2bf(o[2], o[3]);
                        *reinterpret_cast<uint2*>(yp + c) = pk2;
                    }
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + (wm * MT + mt) * 32 + 8 * q + 4 * half;
                    if (co < a.Cout) {                               // Cout % 4 == 0 (checked on the host)
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = acc[mt][nt][4 * q + j] + sbias[co - co0 + j];
                        uint2 pk2;
                        pk2.x = pack2bf(o[0], o[1]);
                        pk2.y = pack2bf(o[2], o[3]);
                        *reinterpret_cast<uint2*>(yp + co) = pk2;
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, int KWT, int PPT>
int conv_launch_t(const Bf16ConvArgs& a, size_t lds, hipStream_t s)
{
    constexpr int BM = WM * MT * 32;
    auto kern = bf16_conv_kernel<WM, WN, MT, NT, KWT, PPT>;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    const unsigned grid = (unsigned)((a.Cout_pad / BM) * a.N * a.tiles_h * a.tiles_w);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kConvThreads), lds, s, a);
    return (int)hipGetLastError();
}

// Register-prefetch depth of the input patch = 16-byte pieces per thread, the smallest instantiated depth that holds it (4: small stride-1
// tiles, 12: 128-pixel stride-2 tiles, 13 / 17: the 512-pixel tiles of the large stride-1 layers stage 816 / 1056 patch pixels, 20: 256-pixel
// stride-2 tiles); anything larger, and run-time kernel widths, use the synchronous staging loop (depth 0).
// (r4, 16 x 512 frames: with 12 the 512-pixel tiles fell onto the synchronous loop -- four dependent rounds of global loads per channel chunk in
//  front of 50 MFMA-bound steps; upSample2 1090 -> 938 us)
int bf16_prefetch_depth(int pieces, int KW, int BM, int BN)
{
    const int need = (pieces + kConvThreads - 1) / kConvThreads;
    if (KW != 5 && KW != 3 && KW != 1 && KW != 6) return 0;
    if (need <= 4) return 4;
    if (KW == 1) return 0;
    if (need <= 12) return 12;
    if (KW == 5 && BM == 128 && BN == 512 && need <= 13) return 13;
    if (KW == 5 && BM == 128 && BN == 512 && need <= 17) return 17;
    if (KW == 5 && BM == 128 && BN == 256 && need <= 20) return 20;
    return 0;
}

template <int WM, int WN, int MT, int NT>
int conv_launch_kw(const Bf16ConvArgs& a, size_t lds, hipStream_t s)
{
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int depth = bf16_prefetch_depth(a.PH * a.PW * 4, a.KW, BM, BN);
    if (a.KW == 5) {
        if (depth == 4) return conv_launch_t<WM, WN, MT, NT, 5, 4>(a, lds, s);
        if (depth == 12) return conv_launch_t<WM, WN, MT, NT, 5, 12>(a, lds, s);
        if constexpr (BM == 128 && BN == 512) {
            if (depth == 13) return conv_launch_t<WM, WN, MT, NT, 5, 13>(a, lds, s);
            if (depth == 17) return conv_launch_t<WM, WN, MT, NT, 5, 17>(a, lds, s);
        }
        if constexpr (BM == 128 && BN == 256) {
            if (depth == 20) return conv_launch_t<WM, WN, MT, NT, 5, 20>(a, lds, s);
        }
        return conv_launch_t<WM, WN, MT, NT, 5, 0>(a, lds, s);
    }
    if (a.KW == 3) {
        if (depth == 4) return conv_launch_t<WM, WN, MT, NT, 3, 4>(a, lds, s);
        if (depth == 12) return conv_launch_t<WM, WN, MT, NT, 3, 12>(a, lds, s);
        return conv_launch_t<WM, WN, MT, NT, 3, 0>(a, lds, s);
    }
    if constexpr (BM == 64 && BN == 64) {          // (the 1-D trunk of the inference forward: three taps x two channel groups, stride 2)
        if (a.KW == 6 && depth == 4) return conv_launch_t<WM, WN, MT, NT, 6, 4>(a, lds, s);
        if (a.KW == 6 && depth == 12) return conv_launch_t<WM, WN, MT, NT, 6, 12>(a, lds, s);
    }
    if (a.KW == 1 && depth == 4) return conv_launch_t<WM, WN, MT, NT, 1, 4>(a, lds, s);
    return conv_launch_t<WM, WN, MT, NT, 0, 0>(a, lds, s);
}

}  // namespace

// tile configuration: 4 = 128 channels x 256 pixels (large stride-1 layers), 0 = 128 channels x 128 pixels, 1 = 64 x 64 (few pixels: more workgroups), 2 = 32 x 128 (few output channels),
// 3 = 128 x 64 (tried for the stride-2 layers -- two workgroups per CU instead of one: measured SLOWER, 187 vs 206 TF/s, so it is
// only reachable through MCVC_BF16_CFG=3 for experiments)
static int conv_config(const Bf16ConvArgs& a)
{
    static const int knob = mcvc_knob("MCVC_BF16_CFG", -1);
    if (a.Cout_pad % 64 != 0) return 2;
    if (knob == 3 && a.stride == 2 && a.Cout_pad % 128 == 0 && !a.glu) return 3;
    const long long px = (long long)a.N * a.OH * a.OW;
    // 128 channels x 256 pixels: a workgroup streams its 128 x KH x KW x Cin weight slice once per PIXEL tile, and on the large stride-1
    // layers that L2 -> LDS weight stream (16 GB per launch at 128 pixels per tile) is the limit: twice the pixels, half the stream
    // 128 channels x 512 pixels, four waves side by side with 4 x 4 accumulators each: 0.5 operand reads (ds_read_b128) per MFMA instead
    // of 0.75 -- at one wave per SIMD the LDS read port, not the matrix pipe, sets the pace of a stage
    if (a.Cout_pad % 128 == 0 && a.stride == 1 && a.KW >= 3 && !a.glu && knob != 0 && knob != 4 && (knob == 5 || (a.Cout_pad / 128) * ((px + 511) / 512) >= 512))
        return 5;
    if (a.Cout_pad % 128 == 0 && a.stride == 1 && a.KW >= 3 && knob != 0 && (knob == 4 || (a.Cout_pad / 128) * ((px + 255) / 256) >= 1024)) return 4;
    // stride-2 layers, 128 channels x 256 pixels: on 128-pixel tiles every (tap, 16-channel) step is 4 operand reads per 4 MFMAs and wave --
    // 128 LDS cycles per 128 matrix cycles and CU before the weight DMA's writes, the LDS port saturated (SQ: MFMA busy 0.31); 64 x 128 per
    // wave is 6 reads per 8 MFMAs.  The 19 x 67-pixel patch (81.5 KB) and the two 40 KB weight stages fit the 160 KB LDS with 256 bytes to spare.
    static const int s2_knob = mcvc_knob("MCVC_BF16_S2_WIDE", 1);
    if (s2_knob && a.stride == 2 && a.KW == 5 && a.KH == 5 && a.Cout_pad % 128 == 0 && !a.glu && (a.Cout_pad / 128) * ((px + 255) / 256) >= 512) {
        int th = 0, twl = 0;
        mcvc_bf16_conv_tile(a.OH, a.OW, a.KH, a.KW, a.stride, 256, &th, &twl);
        if (th > 0) {
            const size_t ph = (size_t)(th - 1) * 2 + a.KH, pw = (size_t)((1 << twl) - 1) * 2 + a.KW;
            if (ph * pw * 4 <= 20 * kConvThreads && (size_t)20 * 4096 + 2 * (size_t)128 * a.KW * 64 <= 160 * 1024) return 4;    // (patch area = 20 prefetch rounds)
        }
    }
    if (a.Cout_pad % 128 != 0 || a.glu == 0) {
        // small problems: 128 x 128 tiles would leave most of the 256 CUs idle
        const long long wg128 = (a.Cout_pad / 128) * ((px + 127) / 128);
        if (a.Cout_pad % 128 != 0 || wg128 < 512) return 1;
    }
    return 0;
}

// the staged patch of a 128-channel tile must leave room for the two weight stages: a stride-2 256-pixel tile has 80 KB = 1280 patch pixels
// (the 4 x 64 shape, fewest tiles on a 20 x 128 grid, needs 1441: the chooser then takes 8 x 32 -- three row tiles for 20 rows, still faster)
static long long bf16_patch_limit(int KW, int stride, int bn) { return (stride == 2 && bn == 256) ? (160 * 1024 - 2 * 128 * KW * 64) / 64 : 96 * 1024 / 64; }

void mcvc_bf16_conv_tile(int OH, int OW, int KH, int KW, int stride, int bn, int* TH, int* tw_log2)
{
    const long long limit = bf16_patch_limit(KW, stride, bn);
    // candidates TH x TW = bn pixels; fewest tiles (least overhang) first, then the smallest staged input patch
    long long best_cost = -1;
    int lmax = 0;
    while ((1 << (lmax + 1)) <= bn) ++lmax;
    for (int l = 3; l <= lmax; ++l) {
        const int tw = 1 << l, th = bn >> l;
        const long long tiles = (long long)cdiv_i(OH, th) * cdiv_i(OW, tw);
        const long long patch = (long long)((th - 1) * stride + KH) * ((tw - 1) * stride + KW);
        if (patch > limit) continue;
        const long long cost = tiles * 4096 + patch;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; *tw_log2 = l; *TH = th; }
    }
}

// LDS bank conflicts of the pixel-operand reads (ds_read_b128) for a patch row pitch of `pw` pixels: the hardware serves a wave's 64
// lanes in four fixed groups of 16 -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table) -- and
// a 32-lane operand spans 32 / TW tile rows, so whether the 16 sixteen-byte slots of a group are distinct depends on the pitch.  Returns the
// extra LDS cycles per group access, averaged over taps / rows / chunk positions (0 = conflict-free).  SQ counters of the round-2 binary
// showed 23 % conflict cycles in the stride-2 layers (16-pixel tile rows, pitch 35); the model gives 6 extra cycles per 4-cycle access
// there and 0 at pitch 40.
static double bf16_patch_conflicts(int tw_log2, int pw, int stride, int KH, int KW)
{
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int TW = 1 << tw_log2, pwe = (pw + 1) >> 1;
    long long extra = 0, n = 0;
    for (int kh = 0; kh < KH; ++kh)
        for (int tap = 0; tap < KW; ++tap)
            for (int lc = 0; lc < 4; ++lc)
                for (int g = 0; g < 2; ++g) {
                    int slot_addr[16][16], cnt[16] = {0};
                    for (int i = 0; i < 16; ++i) {
                        const int l = grp[g][i];
                        const int pr = l >> tw_log2, pc = l & (TW - 1);
                        const int pp = pr * stride * pw + pc + kh * pw + (stride == 2 ? (tap >> 1) + (tap & 1) * pwe : tap);
                        const int addr = (pp << 2) + (lc ^ ((pp >> 2) & 3)), slot = addr & 15;
                        bool dup = false;
                        for (int j = 0; j < cnt[slot]; ++j) dup = dup || slot_addr[slot][j] == addr;
                        if (!dup) slot_addr[slot][cnt[slot]++] = addr;
                    }
                    for (int sl = 0; sl < 16; ++sl) if (cnt[sl] > 1) extra += cnt[sl] - 1;
                    ++n;
                }
    return (double)extra / (double)n;
}

int mcvc_bf16_conv_launch(const Bf16ConvArgs& a0, hipStream_t s)
{
    Bf16ConvArgs a = a0;
    if ((a.Cin & 31) || (a.Cout & 3) || a.KW < 1 || a.KH < 1) return MCVC_ERR_INVALID;
    const int cfg = conv_config(a);
    const int BM = (cfg == 0 || cfg == 3 || cfg == 4 || cfg == 5) ? 128 : (cfg == 1 ? 64 : 32);
    const int BN = (cfg == 5) ? 512 : ((cfg == 4) ? 256 : ((cfg == 1 || cfg == 3) ? 64 : 128));
    if (a.Cout_pad % BM) return MCVC_ERR_INVALID;
    if (a.glu && cfg != 0 && cfg != 4) return MCVC_ERR_INVALID;
    mcvc_bf16_conv_tile(a.OH, a.OW, a.KH, a.KW, a.stride, BN, &a.TH, &a.tw_log2);
    const int TW = 1 << a.tw_log2;
    a.tiles_h = cdiv_i(a.OH, a.TH); a.tiles_w = cdiv_i(a.OW, TW);
    a.PH = (a.TH - 1) * a.stride + a.KH; a.PW = (TW - 1) * a.stride + a.KW;
    {   // widen the staged patch to the nearest row pitch whose operand reads are free of LDS bank conflicts (the extra columns are
        // real image columns or zero fill; nothing reads them)
        static const int knob = mcvc_knob("MCVC_BF16_PITCH", 1);
        if (knob && a.tw_log2 < 5) {
            int best = a.PW; double bc = bf16_patch_conflicts(a.tw_log2, a.PW, a.stride, a.KH, a.KW);
            for (int pw = a.PW + 1; pw <= a.PW + 12 && bc > 0.0; ++pw) {
                if ((size_t)a.PH * pw * 64 > 96 * 1024) break;
                const double c = bf16_patch_conflicts(a.tw_log2, pw, a.stride, a.KH, a.KW);
                if (c < bc) { bc = c; best = pw; }
            }
            a.PW = best;
        }
    }
    if (BM * a.KW * 4 > kMaxWP * kConvThreads) return MCVC_ERR_INVALID;
    // the patch area holds whole prefetch rounds (depth * 256 threads * 16 bytes): every thread stores every piece, no bounds test
    const int depth = bf16_prefetch_depth(a.PH * a.PW * 4, a.KW, BM, BN);
    size_t patch = ((size_t)a.PH * a.PW * 64 + 255) & ~(size_t)255;
    if (patch < (size_t)depth * 4096) patch = (size_t)depth * 4096;
    const size_t wb = (size_t)BM * a.KW * 64;
    a.patch_bytes = (int)patch;
    a.wbufs = 2;                    // weight stages alternate between two LDS buffers (DMA of the next one during the MFMAs)
    const size_t lds = patch + a.wbufs * wb;
    if (lds > 160 * 1024) return MCVC_ERR_INVALID;
    const double px = (double)a.N * a.OH * a.OW;
    TraceScope ts(K_CONV_L, s, 2.0 * px * a.Cout_pad * a.Cin * a.KH * a.KW,
                  2.0 * ((double)a.N * a.H * a.W * a.Cin + px * a.Cout + (double)a.Cout_pad * a.Cin * a.KH * a.KW));
    if (cfg == 0) return conv_launch_kw<2, 2, 2, 2>(a, lds, s);
    if (cfg == 4) return conv_launch_kw<2, 2, 2, 4>(a, lds, s);
    if (cfg == 5) return conv_launch_kw<1, 4, 4, 4>(a, lds, s);
    if (cfg == 1) return conv_launch_kw<2, 2, 1, 1>(a, lds, s);
    if (cfg == 3) return conv_launch_kw<2, 2, 2, 1>(a, lds, s);
    return conv_launch_kw<1, 4, 1, 1>(a, lds, s);
}

// ======================================================================================================================
// InstanceNorm + activation (NHWC bf16)
// ======================================================================================================================
namespace {

// normalised-channel index of conv channel cx: shuffle -> cx / 4, else cx (GLU gate channels: C + c)
// Statistics pass: block = 32 pixel lanes x 8 channel octets (64 conv channels); grid (channel groups, S splits, N).
__global__ void __launch_bounds__(256) bf16_stats_kernel(const Bf16NormArgs a)
{
    __shared__ float red[32 * 64 * 2];
    const int tid = threadIdx.x;
    const int oct = tid & 7, pl = tid >> 3;
    const int c0 = blockIdx.x * 64 + oct * 8;
    const int n = blockIdx.z;
    const int P = a.H * a.W;
    const int per = (P + a.S - 1) / a.S;
    const int p_begin = blockIdx.y * per;
    int p_end = p_begin + per; if (p_end > P) p_end = P;
    float s1[8], s2[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; sh[j] = 0.f; }
    const bool live = c0 < a.Cx;
    if (live) {
        const bf16_t* xb = a.x + (long long)n * a.x_sn + c0;
        {   // common shift of every thread / split for these channels: the first pixel's value (shuffle: of the group's first channel)
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(xb), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) sh[j] = a.shuffle ? f[j & ~3] : f[j];
        }
        for (int p = p_begin + pl; p < p_end; p += 32) {
            const int h = p / a.W, w = p - h * a.W;
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(xb + (long long)h * a.x_sh + (long long)w * a.x_sw), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[j] - sh[j]; s1[j] += d; s2[j] += d * d; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(pl * 64 + oct * 8 + j) * 2] = s1[j]; red[(pl * 64 + oct * 8 + j) * 2 + 1] = s2[j]; }
    __syncthreads();
    if (tid < 128) {
        const int ch = tid >> 1, k = tid & 1;
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[(i * 64 + ch) * 2 + k];
        red[ch * 2 + k] = t;                    // (row 0 of the array is its own destination: each thread only re-writes what it summed)
    }
    __syncthreads();
    // normalised channels of this block: shuffle -> 16 (sum over the 4 conv channels), else 64
    const int Cn = a.shuffle ? a.Cx / 4 : a.Cx;
    const int ncn = a.shuffle ? 16 : 64;
    if (tid < ncn * 2) {
        const int cn_l = tid >> 1, k = tid & 1;
        float t;
        if (a.shuffle) t = red[(4 * cn_l) * 2 + k] + red[(4 * cn_l + 1) * 2 + k] + red[(4 * cn_l + 2) * 2 + k] + red[(4 * cn_l + 3) * 2 + k];
        else t = red[cn_l * 2 + k];
        const int cn = (a.shuffle ? blockIdx.x * 16 : blockIdx.x * 64) + cn_l;
        if (cn < Cn) a.partial[(((long long)n * a.S + blockIdx.y) * Cn + cn) * 2 + k] = t;
    }
}

// stats[n][cn] = (scale, shift) of z = x * scale + shift = gamma * (x - mean) * rstd + beta, from the S partials and the sums' shift
// (re-read from the first pixel).  The affine parameters are folded in HERE, once per (image, channel): the apply pass used to fetch
// gamma / beta / mean / rstd per thread -- 64 scalar loads in front of ~5 pixels of work per thread on the gated layers.
__global__ void __launch_bounds__(256) bf16_finalize_kernel(const Bf16NormArgs a)
{
    const int Cn = a.shuffle ? a.Cx / 4 : a.Cx;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.N * Cn) return;
    const int n = (int)(i / Cn), cn = (int)(i - (long long)n * Cn);
    float s1 = 0.f, s2 = 0.f;
    for (int s = 0; s < a.S; ++s) {
        const float* p = a.partial + (((long long)n * a.S + s) * Cn + cn) * 2;
        s1 += p[0]; s2 += p[1];
    }
    const float cnt = (float)a.H * a.W * (a.shuffle ? 4.f : 1.f);
    const float shift = bf2f(a.x[(long long)n * a.x_sn + (a.shuffle ? 4 * cn : cn)]);
    const float m = s1 / cnt;
    float var = s2 / cnt - m * m;
    if (var < 0.f) var = 0.f;
    const float mean = shift + m, rstd = 1.0f / sqrtf(var + a.eps);
    // channel -> affine parameter: gated layers carry [C value | C gate] channels with their own InstanceNorms (gamma[0] / gamma[1])
    const int Cv = (a.act == BF16_ACT_GLU && !a.shuffle) ? Cn / 2 : Cn;
    const float g = cn < Cv ? a.gamma[0][cn] : a.gamma[1][cn - Cv], b = cn < Cv ? a.beta[0][cn] : a.beta[1][cn - Cv];
    const float sc = rstd * g;
    a.stats[i * 2] = sc;
    a.stats[i * 2 + 1] = b - mean * sc;
}

// one thread = 8 OUTPUT channels (16-byte store) of a run of pixels: the launcher makes the grid stride a multiple of the octet count,
// so a thread keeps its channel octet and its statistics / affine parameters stay in registers while it walks over pixels
// (they are re-read only when it crosses into the next image).  shuffle: 32 conv channels in, 4 output pixels.
__global__ void __launch_bounds__(256) bf16_apply_kernel(const Bf16NormArgs a)
{
    const int C = a.shuffle ? a.Cx / 4 : (a.act == BF16_ACT_GLU ? a.Cx / 2 : a.Cx);      // output channels
    const int noct = C >> 3;
    const int P = a.H * a.W;
    // grid (x, n): one image per blockIdx.y, 32-bit index arithmetic (the 64-bit divisions of a flat (n, pixel, octet) index made this
    // pass ALU-bound at ~1.5 TB/s); the launcher makes the x stride a multiple of the octet count, so a thread keeps its channel octet
    const int n = blockIdx.y;
    const int stride = (int)gridDim.x * 256;
    const int pstep = stride / noct;
    int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int c0 = (idx % noct) * 8;
    const int Cs = (a.act == BF16_ACT_GLU) ? 2 * C : C;                                    // channels of the statistics table
    float sc0[8], sh0[8], sc1[8], sh1[8];                                                  // z = x * sc + sh
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc0[j] = 1.f; sh0[j] = 0.f; sc1[j] = 1.f; sh1[j] = 0.f; }
    if (a.has_norm) {        // (scale, shift) pairs of 8 consecutive channels = 64 contiguous bytes of the finalize pass's table (c0 % 8 == 0: 16-byte aligned)
        const float4* st = reinterpret_cast<const float4*>(a.stats + ((long long)n * Cs + c0) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 v = st[q]; sc0[2 * q] = v.x; sh0[2 * q] = v.y; sc0[2 * q + 1] = v.z; sh0[2 * q + 1] = v.w; }
        if (a.act == BF16_ACT_GLU) {
            const float4* sg = reinterpret_cast<const float4*>(a.stats + ((long long)n * Cs + C + c0) * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = sg[q]; sc1[2 * q] = v.x; sh1[2 * q] = v.y; sc1[2 * q + 1] = v.z; sh1[2 * q + 1] = v.w; }
        }
    }
    for (int p = idx / noct; p < P; p += pstep) {
        const int h = p / a.W, w = p - h * a.W;
        const bf16_t* xp = a.x + (long long)n * a.x_sn + (long long)h * a.x_sh + (long long)w * a.x_sw;
        auto out_ptr = [&](int oh, int ow) {
            long long o = (long long)n * a.y_sn + (long long)oh * a.y_sh + (long long)ow * a.y_sw;
            if (a.y_csplit > 0) o += (long long)(c0 / a.y_csplit) * a.y_sc2 + (c0 % a.y_csplit); else o += c0;
            return o;
        };
        if (a.shuffle) {
            float f[32];
#pragma unroll
            for (int q = 0; q < 4; ++q) unpack8(*reinterpret_cast<const uint4*>(xp + 4 * c0 + 8 * q), f + 8 * q);
#pragma unroll
            for (int ij = 0; ij < 4; ++ij) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = f[4 * j + ij] * sc0[j] + sh0[j];
                    o[j] = (a.act == BF16_ACT_SILU) ? z * sigmoidf_(z) : z;
                }
                *reinterpret_cast<uint4*>(a.y + out_ptr(2 * h + (ij >> 1), 2 * w + (ij & 1))) = pack8(o);
            }
        } else {
            float f[8], o[8];
            unpack8(*reinterpret_cast<const uint4*>(xp + c0), f);
            if (a.act == BF16_ACT_GLU) {
                float fg[8];
                unpack8(*reinterpret_cast<const uint4*>(xp + C + c0), fg);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f[j] * sc0[j] + sh0[j]) * sigmoidf_(fg[j] * sc1[j] + sh1[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = f[j] * sc0[j] + sh0[j];
                    o[j] = (a.act == BF16_ACT_SILU) ? z * sigmoidf_(z) : z;
                }
            }
            const long long yo = out_ptr(h, w);
            if (a.res) {
                float r[8];
                unpack8(*reinterpret_cast<const uint4*>(a.res + yo), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            *reinterpret_cast<uint4*>(a.y + yo) = pack8(o);
        }
    }
}

// Small planes (the 1-D trunk at inference: one row of T/4 <= 32 * PPT pixels per sample): statistics, normalisation, activation / GLU /
// residual in ONE launch -- a workgroup owns 64 output channels (GLU: + their 64 gate channels) of one sample and keeps them in registers
// between the two passes.  The three-launch form (statistics over pixel splits, finalize, apply) is launch-latency for these layers:
// 3 x ~6 us around a 17-25 us convolution, 13 times per forward.  Same arithmetic (sums shifted by the first pixel, fp32).
template <int PPT>
__global__ void __launch_bounds__(256) bf16_norm_small_kernel(const Bf16NormArgs a)
{
    __shared__ float red[32 * 128 * 2];
    __shared__ float stat[128 * 2];                          // (mean, rstd) of the 64 value channels, then of the 64 gate channels
    const int tid = threadIdx.x;
    const int oct = tid & 7, pl = tid >> 3;
    const bool glu = a.act == BF16_ACT_GLU;
    const int C = glu ? a.Cx / 2 : a.Cx;                     // output channels
    const int c0 = blockIdx.x * 64 + oct * 8;
    const int n = blockIdx.y;
    const int P = a.H * a.W;
    const bool live = c0 < C;
    const bf16_t* xb = a.x + (long long)n * a.x_sn;
    float f[PPT][8], g[PPT][8], sh[8], shg[8];
    float s1[8], s2[8], t1[8], t2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = s2[j] = t1[j] = t2[j] = 0.f; sh[j] = shg[j] = 0.f; }
    if (live) {
        unpack8(*reinterpret_cast<const uint4*>(xb + c0), sh);
        if (glu) unpack8(*reinterpret_cast<const uint4*>(xb + C + c0), shg);
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = pl + 32 * i;
        const bool ok = live && p < P;
        const int h = ok ? p / a.W : 0, w = ok ? p - h * a.W : 0;
        const bf16_t* xp = xb + (long long)h * a.x_sh + (long long)w * a.x_sw;
        if (ok) { unpack8(*reinterpret_cast<const uint4*>(xp + c0), f[i]); if (glu) unpack8(*reinterpret_cast<const uint4*>(xp + C + c0), g[i]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!ok) { f[i][j] = sh[j]; g[i][j] = shg[j]; }
            const float d = f[i][j] - sh[j]; s1[j] += d; s2[j] += d * d;
            if (glu) { const float e = g[i][j] - shg[j]; t1[j] += e; t2[j] += e * e; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[(pl * 128 + oct * 8 + j) * 2] = s1[j]; red[(pl * 128 + oct * 8 + j) * 2 + 1] = s2[j];
        red[(pl * 128 + 64 + oct * 8 + j) * 2] = t1[j]; red[(pl * 128 + 64 + oct * 8 + j) * 2 + 1] = t2[j];
    }
    __syncthreads();
    {   // 256 threads = 128 channels x (sum, sum of squares)
        const int ch = tid >> 1, k = tid & 1;
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[(i * 128 + ch) * 2 + k];
        red[ch * 2 + k] = t;                    // (row 0 is its own destination: each thread only re-writes what it summed)
    }
    __syncthreads();
    if (tid < 128) {
        const int ch = tid;                     // 0..63 value, 64..127 gate
        const int cx = (ch < 64) ? blockIdx.x * 64 + ch : C + blockIdx.x * 64 + (ch - 64);
        float m = 0.f, r = 1.f;
        if ((ch < 64 || glu) && (blockIdx.x * 64 + (ch & 63)) < C) {
            const float cnt = (float)P;
            const float shift = bf2f(xb[cx]);
            const float mm = red[ch * 2] / cnt;
            float var = red[ch * 2 + 1] / cnt - mm * mm;
            if (var < 0.f) var = 0.f;
            m = shift + mm; r = 1.0f / sqrtf(var + a.eps);
        }
        stat[ch * 2] = m; stat[ch * 2 + 1] = r;
    }
    __syncthreads();
    if (!live) return;
    float sc0[8], sh0[8], sc1[8], sh1[8];                    // z = x * sc + sh
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = oct * 8 + j;
        const float g0 = a.gamma[0][c0 + j], b0 = a.beta[0][c0 + j];
        sc0[j] = stat[cl * 2 + 1] * g0; sh0[j] = b0 - stat[cl * 2] * sc0[j];
        sc1[j] = 1.f; sh1[j] = 0.f;
        if (glu) { const float g1 = a.gamma[1][c0 + j], b1 = a.beta[1][c0 + j]; sc1[j] = stat[(64 + cl) * 2 + 1] * g1; sh1[j] = b1 - stat[(64 + cl) * 2] * sc1[j]; }
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = pl + 32 * i;
        if (p >= P) continue;
        const int h = p / a.W, w = p - h * a.W;
        long long yo = (long long)n * a.y_sn + (long long)h * a.y_sh + (long long)w * a.y_sw;
        if (a.y_csplit > 0) yo += (long long)(c0 / a.y_csplit) * a.y_sc2 + (c0 % a.y_csplit); else yo += c0;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = f[i][j] * sc0[j] + sh0[j];
            if (glu) o[j] = z * sigmoidf_(g[i][j] * sc1[j] + sh1[j]);
            else o[j] = (a.act == BF16_ACT_SILU) ? z * sigmoidf_(z) : z;
        }
        if (a.res) {
            float r[8];
            unpack8(*reinterpret_cast<const uint4*>(a.res + yo), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += r[j];
        }
        *reinterpret_cast<uint4*>(a.y + yo) = pack8(o);
    }
}

}  // namespace

int mcvc_bf16_norm_splits(int N, int P, int Cn)
{
    // enough workgroups to fill the chip, at least 64 pixels per split
    const int groups = N * cdiv_i(Cn, 64);
    int S = cdiv_i(1024, groups > 0 ? groups : 1);
    const int cap = cdiv_i(P, 64);
    if (S > cap) S = cap;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    return S;
}

long long mcvc_bf16_norm_partial_floats(const Bf16NormArgs& a)
{
    const int Cn = a.shuffle ? a.Cx / 4 : a.Cx;
    return (long long)a.N * a.S * Cn * 2;
}

int mcvc_bf16_norm_launch(const Bf16NormArgs& a, hipStream_t s)
{
    if ((a.Cx & 7) || (a.shuffle && (a.Cx & 31))) return MCVC_ERR_INVALID;
    const int C = a.shuffle ? a.Cx / 4 : (a.act == BF16_ACT_GLU ? a.Cx / 2 : a.Cx);
    if (C & 7) return MCVC_ERR_INVALID;
    const double el = (double)a.N * a.H * a.W * a.Cx;
    static const int small_knob = mcvc_knob("MCVC_BF16_NORM_SMALL", 1);
    if (small_knob && a.has_norm && !a.shuffle && a.H * a.W <= 256 && (C % 8) == 0) {
        TraceScope ts(K_NORM_FWD, s, 0.0, 2.0 * (el + (double)a.N * a.H * a.W * C));
        const dim3 grid((unsigned)cdiv_i(C, 64), (unsigned)a.N);
        if (a.H * a.W <= 128) hipLaunchKernelGGL(bf16_norm_small_kernel<4>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(bf16_norm_small_kernel<8>, grid, dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.has_norm) {
        if (a.S < 1) return MCVC_ERR_INVALID;
        {
            TraceScope ts(K_NORM_FWD, s, 0.0, 2.0 * el);
            hipLaunchKernelGGL(bf16_stats_kernel, dim3((unsigned)cdiv_i(a.Cx, 64), (unsigned)a.S, (unsigned)a.N), dim3(256), 0, s, a);
        }
        const int Cn = a.shuffle ? a.Cx / 4 : a.Cx;
        hipLaunchKernelGGL(bf16_finalize_kernel, dim3((unsigned)cdiv_ll((long long)a.N * Cn, 256)), dim3(256), 0, s, a);
    }
    const long long work = (long long)a.H * a.W * (C >> 3);           // per image (grid.y = image)
    long long blocks = cdiv_ll(work, 256 * 4);              // ~4 pixels per thread
    const long long cap = 4096 / (a.N < 1 ? 1 : a.N) + 1;
    if (blocks > cap) blocks = cap;
    {   // grid stride a multiple of the octet count: a thread keeps its channel octet (statistics stay in registers)
        const int noct = C >> 3;
        int g = noct, r = 256;
        while (r) { const int t = g % r; g = r; r = t; }    // gcd(noct, 256)
        const int unit = noct / g;
        blocks = cdiv_ll(blocks, unit) * unit;
    }
    TraceScope ts(K_NORM_FWD, s, 0.0, 2.0 * (el + (double)a.N * a.H * a.W * C * (a.shuffle ? 4.0 : 1.0)));
    hipLaunchKernelGGL(bf16_apply_kernel, dim3((unsigned)blocks, (unsigned)a.N), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// ======================================================================================================================
// edges
// ======================================================================================================================
namespace {

__global__ void __launch_bounds__(256) bf16_prep_kernel(const float* __restrict__ x, const float* __restrict__ mask, bf16_t* __restrict__ xin,
                                                        int B, int H, int W)
{
    // one thread = one (pixel, octet of the 32 folded channels): channels kw*2 + ci, kw = 4*oct .. 4*oct+3
    const long long total = (long long)B * H * W * 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int oct = (int)(idx & 3);
        const long long pix = idx >> 2;
        const int w = (int)(pix % W);
        const long long row = pix / W;              // b*H + h
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kw = oct * 4 + j;
            const int iw = w + kw - 7;
            float xv = 0.f, mv = 0.f;
            if (kw < 15 && iw >= 0 && iw < W) {
                mv = mask ? mask[row * W + iw] : 1.0f;
                xv = x[row * W + iw] * mv;
            }
            o[2 * j] = xv; o[2 * j + 1] = mv;
        }
        *reinterpret_cast<uint4*>(xin + pix * 32 + oct * 8) = pack8(o);
    }
}

// ---- conv1 of the inference forward, fused with its input preparation and its gated GLU (r6) -----------------------------------------
// model.py:241-242  conv1(stack(x * mask, mask)) * sigmoid(conv1_gates(...)),  2 -> 128 | 128 channels, 5 x 15, padding (2, 7).
// As a generic tile (bf16_conv_kernel<2,2,2,2,1,4> over the folded input of bf16_prep_kernel) the layer was 10 240 workgroups of FIVE
// pipeline stages each -- prologue, weight DMA and epilogue latency around 10 MFMA steps -- and ran at 1.6 TB/s, a third of what its
// 210 MB of traffic allow (133 + 16 us of a 2.93 ms forward).  Here a workgroup owns a strip of 32 output columns x SH rows of one sample:
//   * the strip's (x * mask, mask) pairs with halo -- (SH + 4) rows x 46 columns, 4 bytes per pixel -- are built ONCE in LDS straight from
//     the fp32 inputs: the folded 32-channel tensor (42 MB written + 42 MB read per forward) and its kernel no longer exist;
//   * the layer's whole weight slice of a wave (64 packed rows = 32 value + their 32 gate channels, K = 5 x 32) lives in REGISTERS
//     (2 x 10 x 4 VGPRs), read once per workgroup; the four waves split the 128 output channels and share the staged strip;
//   * per output row: 10 k-steps x 2 MFMAs per wave, the pixel operand of a k-step = 4 consecutive staged pixels (kw = 8 h2 + 4 half + j)
//     of row r + kh; accumulators start from the bias; GLU + bf16 store as in the generic epilogue.
// Same arithmetic as prep + conv (bf16 products of the same operands, fp32 accumulation; only the summation order inside the MFMA chain
// differs: kh-major here as there).
constexpr int kC1W = 32, kC1Halo = 14, kC1Pitch = kC1W + kC1Halo + 1;      // staged row pitch 47 words (odd: the two lane halves' windows spread over the banks)
template <int SH>
__global__ void __launch_bounds__(256) bf16_conv1_fused_kernel(const float* __restrict__ x, const float* __restrict__ mask, const bf16_t* __restrict__ w,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ y, int H, int W, int segs)
{
    __shared__ unsigned xs[(SH + 4) * kC1Pitch];
    __shared__ __attribute__((aligned(16))) float sbias[256];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int strips = (W + kC1W - 1) / kC1W;
    int b = blockIdx.x;
    const int strip = b % strips; b /= strips;
    const int seg = b % segs; const int n = b / segs;
    const int w0 = strip * kC1W, h0 = seg * SH;
    // ---- weights -> registers: packed [256][5][32] bf16, rows in 64-row blocks [32 value | 32 gate] (Bf16PackArgs::glu_interleave)
    u32x4 areg[2][10];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const bf16_t* wr = w + (long long)(wave * 64 + mt * 32 + l31) * 160 + 8 * half;
#pragma unroll
        for (int s = 0; s < 10; ++s) areg[mt][s] = *reinterpret_cast<const u32x4*>(wr + (s >> 1) * 32 + (s & 1) * 16);
    }
    sbias[tid] = bias ? bias[tid] : 0.f;
    // ---- the strip: staged column c <-> image column w0 - 7 + c, staged row r <-> image row h0 - 2 + r; zero outside the image
    const float* xn = x + (long long)n * H * W;
    const float* mn = mask ? mask + (long long)n * H * W : nullptr;
    for (int i = tid; i < (SH + 4) * kC1Pitch; i += 256) {       // (all 47 columns: the last one is read by the zero-weight tap kw = 15 and must be finite)
        const int r = i / kC1Pitch, c = i - r * kC1Pitch;
        const int ih = h0 - 2 + r, iw = w0 - 7 + c;
        float xv = 0.f, mv = 0.f;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
            mv = mn ? mn[(long long)ih * W + iw] : 1.0f;
            xv = xn[(long long)ih * W + iw] * mv;
        }
        xs[r * kC1Pitch + c] = pack2bf(xv, mv);
    }
    __syncthreads();
    const int ow = w0 + l31;
    const unsigned* xl = xs + l31 + 4 * half;
    for (int r = 0; r < SH; ++r) {
        const int oh = h0 + r;
        if (oh >= H) break;
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(sbias + wave * 64 + mt * 32 + 8 * q + 4 * half);
                acc[mt][4 * q] = bv.x; acc[mt][4 * q + 1] = bv.y; acc[mt][4 * q + 2] = bv.z; acc[mt][4 * q + 3] = bv.w;
            }
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const unsigned* p = xl + (r + (s >> 1)) * kC1Pitch + 8 * (s & 1);
            u32x4 bv;
            bv.x = p[0]; bv.y = p[1]; bv.z = p[2]; bv.w = p[3];
            const bf16x8 bb = __builtin_bit_cast(bf16x8, bv);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[0][s]), bb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[1][s]), bb, acc[1], 0, 0, 0);
        }
        if (ow < W) {
            bf16_t* yp = y + (((long long)n * H + oh) * W + ow) * 128 + wave * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[0][4 * q + j] * sigmoidf_(acc[1][4 * q + j]);
                uint2 pk2;
                pk2.x = pack2bf(o[0], o[1]);
                pk2.y = pack2bf(o[2], o[3]);
                *reinterpret_cast<uint2*>(yp + 8 * q + 4 * half) = pk2;
            }
        }
    }
}

__global__ void __launch_bounds__(256) bf16_last_kernel(const bf16_t* __restrict__ z, const float* __restrict__ bias, float* __restrict__ out,
                                                        int B, int H, int W)
{
    const long long total = (long long)B * H * W;
    const float b = bias ? bias[0] : 0.f;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
        const int w = (int)(pix % W);
        const long long row = pix / W;
        float s = b;
#pragma unroll
        for (int kw = 0; kw < 15; ++kw) {
            const int iw = w + kw - 7;
            if (iw >= 0 && iw < W) s += bf2f(z[(row * W + iw) * 32 + kw]);
        }
        out[pix] = s;
    }
}

// ---- the last conv of the inference forward (128 -> 1, 5 x 15, padding (2, 7): model.py:207-211, 278-279) in ONE launch (r6) -----------
// The generic form computed the 15 kernel columns as 15 (of 32) output channels of a 5 x 1 conv -- z[b][h][w][32] bf16, 42 MB written and
// re-read -- and a second kernel summed the shifted planes (73.8 + 34.5 us; the 168 MB activation read allows ~35).  Here:
//   * Z[kw][w'] = sum_{kh, ci} W[ci][kh][kw] x[ci][h + kh - 2][w'] per INPUT column w' on v_mfma_f32_16x16x32_bf16 (16 rows = 15 taps + a zero
//     row: no wasted half tile), the whole weight tensor (16 x 640 bf16) in REGISTERS (5 x 4 operands), the pixel operand straight from
//     global memory (a lane's 8 k = 8 consecutive channels of its pixel: one 16-byte load), next row's loads in flight during this row's MFMAs;
//   * every input row is loaded ONCE and feeds the five output rows it reaches (kh = 0..4) -- five rotating accumulator sets, the row loop
//     unrolled by five so that the rotation is a compile-time renaming; an output row is complete after its kh = 4 contribution;
//   * y[h][w] = bias + sum_kw Z[kw][w + kw - 7]: the finished 16 x 128 fp32 tile goes through LDS (double-buffered: one barrier per row),
//     114 threads sum their anti-diagonal.  fp32 all the way (the two-kernel form rounded Z to bf16 in between).
constexpr int kLastCols = 128, kLastOut = kLastCols - 14, kLastPitch = kLastCols + 4;
template <int SH>
__global__ void __launch_bounds__(256) bf16_last_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, int H, int W, int segs)
{
    __shared__ float Zs[2][16][kLastPitch];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int strips = (W + kLastOut - 1) / kLastOut;
    int b = blockIdx.x;
    const int strip = b % strips; b /= strips;
    const int seg = b % segs; const int n = b / segs;
    const int c0 = strip * kLastOut - 7, h0 = seg * SH;
    // weights: packed [32][5][4][32] bf16 (BF16_PACK_KW_OUT: row = kernel column, rows >= 15 zero); lane: row l15, k = 8 kg .. 8 kg + 7 of a 32-k step
    u32x4 areg[5][4];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) areg[kh][cc] = *reinterpret_cast<const u32x4*>(w + ((l15 * 5 + kh) * 4 + cc) * 32 + 8 * kg);
    const float bs = bias ? bias[0] : 0.f;
    int col[2]; bool cok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { col[nt] = c0 + wave * 32 + nt * 16 + l15; cok[nt] = col[nt] >= 0 && col[nt] < W; }
    // every load is unconditional (row / column clamped into the image, the value zeroed by a select where it lies outside): a conditional load
    // ends the compiler's vmcnt bookkeeping and it then drains the queue after each one (r6: the first form had 50 `vmcnt(0)` in its row loop)
    auto load_row = [&](int i, u32x4 (&bv)[2][4]) __attribute__((always_inline)) {
        int ih = h0 - 2 + i;
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bf16_t* p = x + (((long long)n * H + ih) * W + (cok[nt] ? col[nt] : 0)) * 128 + 8 * kg;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) bv[nt][cc] = *reinterpret_cast<const u32x4*>(p + cc * 32);
        }
    };
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ acc[5][2];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[s][nt] = f32x4_{0.f, 0.f, 0.f, 0.f};
    // rows fully unrolled (SH + 4 input rows): the five rotating accumulator sets and the three-deep operand ring are compile-time
    // renamings, no register is copied at a loop's back edge (which would wait for the loads in flight); two rows of loads stay in flight
    constexpr int NR = SH + 4;
    u32x4 bq[3][2][4];
    load_row(0, bq[0]);
    load_row(1, bq[1]);
    int buf = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        if (i + 2 < NR) load_row(i + 2, bq[(i + 2) % 3]);
        const int ih = h0 - 2 + i;
        const bool rok = ih >= 0 && ih < H;                          // (uniform)
        u32x4 bm[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) bm[nt][cc] = (rok && cok[nt]) ? bq[i % 3][nt][cc] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
            const int slot = (i + 10 - kh) % 5;                      // output row i - kh
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const f32x4_ cin = (kh == 0 && cc == 0) ? f32x4_{0.f, 0.f, 0.f, 0.f} : acc[slot][nt];       // kh = 0 opens the output row
                    acc[slot][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, areg[kh][cc]),
                                                                            __builtin_bit_cast(bf16x8, bm[nt][cc]), cin, 0, 0, 0);
                }
        }
        const int r = i - 4;                                         // the output row whose kh = 4 contribution was just added
        if (r >= 0 && r < SH && h0 + r < H) {                        // (compile-time && uniform)
            const int slot = (i + 1) % 5;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) Zs[buf][4 * kg + j][wave * 32 + nt * 16 + l15] = acc[slot][nt][j];
            __syncthreads();
            const int ow = strip * kLastOut + tid;
            if (tid < kLastOut && ow < W) {
                float s_ = bs;
#pragma unroll
                for (int kw = 0; kw < 15; ++kw) s_ += Zs[buf][kw][tid + kw];
                out[((long long)n * H + h0 + r) * W + ow] = s_;
            }
            buf ^= 1;
        }
    }
}

// ---- a residual-block layer of the inference forward in ONE launch (r6) ---------------------------------------------------------------
// model.py:47-76 at inference: Conv1d(k = 3, padding 1) + InstanceNorm1d(affine) + {value * sigmoid(gate) | + residual}.  Until r6 each of the
// 12 layers was a 64 x 64-tile convolution (8-16 barrier-bound stages, 14.3 us) + a one-launch InstanceNorm (9.7 us) that re-read the
// conv output.  InstanceNorm needs a whole (sample, channel) row; at T <= 512 frames that row is W = T/4 <= 128 columns = ONE MFMA tile
// width, so the convolution's own workgroup can finish the layer:
//   * a workgroup owns 32 output channels of ONE sample across all W columns.  GLU layers: 64 MFMA rows = per 32-row tile [16 value | their
//     16 gate] rows -- accumulator registers r and r + 8 of a lane are then the value and the gate of the SAME channel and column, the GLU
//     needs no exchange; waves 2 (row tiles) x 2 (column halves).  Plain layers: one 32-row tile, waves 1 x 4.
//   * the sample's input row block [W + 2][Cin] (zero halo) is staged ONCE in LDS (pixel pitch Cin * 2 + 16 bytes: the 32 pixels of an
//     operand read hit distinct 16-byte bank slots); tap kw reads the window shifted by kw pixels.
//   * the weights are packed in operand order -- [row tile][k-step][64 lanes][8 bf16], k = kw * Cin + ci -- so a wave's A operand of a
//     k-step is ONE coalesced 1 KB load straight into registers (ring of eight steps in flight; no LDS, no barrier in the K loop).
//   * epilogue in registers: per-row sum / sum of squares over the valid columns (lane butterfly + one LDS hand-over between the waves that
//     share rows), scale / shift from gamma, beta (the conv bias cancels under the norm and is never read), GLU or residual, bf16 store.
// Statistics are taken on the fp32 accumulators (the two-launch form took them on the bf16-rounded conv output).
struct Bf16TrunkArgs {
    const bf16_t* x; long long x_sn;          // [B][W][Cin]
    const bf16_t* w;                          // [row tile][k-step][64][8] (bf16_trunk_pack_kernel)
    const float* g0; const float* b0; const float* g1; const float* b1;       // affine of the value (| gate) InstanceNorm, [C]
    const bf16_t* res;                        // residual [B][W][C] or null
    bf16_t* y; long long y_sn;                // [B][W][C]
    int W, Cin, C;
    float eps;
};
constexpr int kTrunkMaxW = 128;

template <bool GLU, int CIN>
__global__ void __launch_bounds__(256) bf16_trunk_layer_kernel(const Bf16TrunkArgs a)
{
    // (CIN is a template parameter so that the K loop unrolls COMPLETELY: with a loop back edge the compiler copies the weight ring's registers
    //  there and waits vmcnt(0) for them -- the ring then drained every eight steps, 12 L2 round trips per layer)
    constexpr int WN = GLU ? 2 : 4, NT = GLU ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = GLU ? (wave >> 1) : 0, wn = GLU ? (wave & 1) : wave;
    const int l31 = lane & 31, half = lane >> 5;
    const int n = blockIdx.y;
    const int wgt = blockIdx.x;                                  // 32 output channels
    const int tile = GLU ? (wgt * 2 + wm) : wgt;                 // 32-row tile of the packed weights
    constexpr int S = 3 * CIN / 16;                              // k-steps (k = kw * Cin + ci)
    constexpr int pitch = CIN * 2 + 16;                          // bytes per staged pixel
    unsigned char* Xs = smem;                                    // [kTrunkMaxW + 2][pitch]
    float* red = reinterpret_cast<float*>(smem + (size_t)(kTrunkMaxW + 2) * pitch);      // [2][WN][32][2]
    // ---- A ring: the first eight k-steps are requested before anything else
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + (long long)tile * S * 64 + lane;
    u32x4 aq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) aq[u] = wp[(long long)u * 64];
    // ---- stage the sample's rows: staged pixel p' <-> image pixel p' - 1; zero outside [0, W)
    {
        constexpr int c8n = CIN >> 3;                            // 16-byte pieces per pixel
        const int total = (kTrunkMaxW + 2) * c8n;
        const bf16_t* xb = a.x + (long long)n * a.x_sn;
        // (17 pieces per thread in flight at once: the whole 256-channel block in one round of loads, the 512-channel block in two --
        //  eight at a time made this staging five dependent L2 round trips, 10 us of a 46 us layer)
        constexpr int SB = 17;
        for (int i0 = tid; i0 < total; i0 += 256 * SB) {
            u32x4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = i0 + u * 256;
                const int pp = i / c8n, c8 = i - pp * c8n, px = pp - 1;
                v[u] = u32x4{0u, 0u, 0u, 0u};
                if (i < total && px >= 0 && px < a.W) v[u] = *reinterpret_cast<const u32x4*>(xb + (long long)px * CIN + c8 * 8);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = i0 + u * 256;
                const int pp = i / c8n, c8 = i - pp * c8n;
                if (i < total) *reinterpret_cast<u32x4*>(Xs + (size_t)pp * pitch + c8 * 16) = v[u];
            }
        }
    }
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int col0 = wn * (32 * NT) + l31;                       // this lane's column of n-tile 0
    const unsigned xs0 = lds_addr(Xs) + (unsigned)col0 * (unsigned)pitch + (unsigned)half * 16u;
    // K loop: the pixel operand of step s + 1 is requested BEFORE step s waits (LDS returns in order: lgkmcnt(NT) = "everything but the newest NT
    // reads has landed"), so a step costs its MFMAs, not an LDS round trip; two operand sets alternate by the step's parity (eight steps per
    // block: the parity is a compile-time property of the unrolled body, nothing is copied at the loop's back edge)
    auto blk_addr = [&](int s0) __attribute__((always_inline)) {
        const int k0 = s0 * 16, kw = k0 / CIN, ci0 = k0 - kw * CIN;                // (eight steps = 128 channels of one tap: Cin % 128 == 0)
        return xs0 + (unsigned)kw * (unsigned)pitch + (unsigned)ci0 * 2u;
    };
    u32x4 bv[2][NT];
    auto request = [&](unsigned base, int u, u32x4 (&b_)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(b_[nt]) : "v"(base + (unsigned)(nt * 32) * (unsigned)pitch), "i"(u * 32));
    };
    unsigned xb_ = blk_addr(0);
    request(xb_, 0, bv[0]);
#pragma unroll
    for (int s0 = 0; s0 < S; s0 += 8) {
        const unsigned xnext = blk_addr(s0 + 8 < S ? s0 + 8 : s0);
        // (branch-free body: past the end the ring re-requests the last step's weights and the first operand of this block -- a conditional
        //  load ends the compiler's counter tracking, and it then waits vmcnt(0) after every weight load: the ring overlapped nothing)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < 7) request(xb_, u + 1, bv[(u + 1) & 1]);
            else request(xnext, 0, bv[0]);
            const u32x4 av = aq[u];
            const int sn = (s0 + 8 + u < S) ? (s0 + 8 + u) : (S - 1);
            aq[u] = wp[(long long)sn * 64];
            if constexpr (NT == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bv[u & 1][0]), "+v"(bv[u & 1][1]));
            else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bv[u & 1][0]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv[u & 1][nt]), acc[nt], 0, 0, 0);
        }
        xb_ = xnext;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the dangling request of the last step)
    // ---- statistics of every MFMA row over the valid columns.  Register r <-> row (r & 3) + 8 (r >> 2) + 4 half.
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const bool ok = col0 + nt * 32 < a.W;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = ok ? acc[nt][r] : 0.f; s1[r] += v; s2[r] += v * v; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* q = red + (((wm * WN + wn) * 32 + row) << 1);
            q[0] = s1[r]; q[1] = s2[r];
        }
    }
    __syncthreads();
    const float invW = 1.0f / (float)a.W;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) { const float* q = red + (((wm * WN + j) * 32 + row) << 1); t1 += q[0]; t2 += q[1]; }
        const float mean = t1 * invW;
        float var = t2 * invW - mean * mean;
        if (var < 0.f) var = 0.f;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        float g, b;
        if (GLU) { const int c = tile * 16 + (row & 15); g = (row < 16) ? a.g0[c] : a.g1[c]; b = (row < 16) ? a.b0[c] : a.b1[c]; }
        else { const int c = tile * 32 + row; g = a.g0[c]; b = a.b0[c]; }
        sc[r] = rstd * g; sh[r] = b - mean * sc[r];
    }
    // ---- normalise, GLU / residual, store (4 consecutive channels per register quad: 8-byte stores)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int px = col0 + nt * 32;
        if (px >= a.W) continue;
        const long long po = (long long)n * a.y_sn + (long long)px * a.C;
        if constexpr (GLU) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * q + j;
                    const float z0 = acc[nt][r] * sc[r] + sh[r], z1 = acc[nt][r + 8] * sc[r + 8] + sh[r + 8];
                    o[j] = z0 * sigmoidf_(z1);
                }
                uint2 pk2; pk2.x = pack2bf(o[0], o[1]); pk2.y = pack2bf(o[2], o[3]);
                *reinterpret_cast<uint2*>(a.y + po + tile * 16 + 8 * q + 4 * half) = pk2;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = tile * 32 + 8 * q + 4 * half;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[nt][4 * q + j] * sc[4 * q + j] + sh[4 * q + j];
                if (a.res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(a.res + po + c);
                    o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
                    o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
                }
                uint2 pk2; pk2.x = pack2bf(o[0], o[1]); pk2.y = pack2bf(o[2], o[3]);
                *reinterpret_cast<uint2*>(a.y + po + c) = pk2;
            }
        }
    }
}

// weights of a trunk layer in operand order: dst[((tile * S + s) * 64 + lane) * 8 + j] = W[row(tile, lane & 31)][k = 16 s + 8 (lane >> 5) + j],
// k = kw * Cin + ci; GLU: tile rows 0-15 = value channels 16 tile .. 16 tile + 15 (w0), rows 16-31 = their gates (w1); source OIHW [Cout][Cin][1][3]
__global__ void __launch_bounds__(256) bf16_trunk_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1, bf16_t* __restrict__ dst,
                                                              int Cin, int tiles, int glu)
{
    const int S = 3 * Cin / 16;
    const long long total = (long long)tiles * S * 64 * 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long ts = idx >> 9;
        const int s = (int)(ts % S), tile = (int)(ts / S);
        const int row = lane & 31, k = 16 * s + 8 * (lane >> 5) + j;
        const int kw = k / Cin, ci = k - kw * Cin;
        const float* src = w0; int co;
        if (glu) { co = tile * 16 + (row & 15); if (row >= 16) src = w1; }
        else co = tile * 32 + row;
        dst[idx] = f2bf(src[((long long)co * Cin + ci) * 3 + kw]);
    }
}

// ---- conv2dto1d of the inference forward + its InstanceNorm in ONE launch (r6) ---------------------------------------------------------
// model.py:142-146, 254-255: Conv1d(5120 -> 256, k = 1) + InstanceNorm1d over a sample's W = T/4 <= 128 columns.  As a 64 x 64-tile convolution
// (1 x 5 stride 5 over 1024-channel "pixels") the layer was 32 barrier-bound stages per workgroup on 128 of the 256 compute units: 73 us for
// 5.4 GF, + 8 us for the norm.  Here a workgroup owns 32 output channels x all columns of one sample (the norm's row is in the tile):
//   * X [W][5120] streams through a FOUR-stage LDS-DMA ring in chunks of 128 channels (32 KB: 128 pixel rows of 256 bytes; piece p of row r at
//     slot p ^ (r & 15) -- the DMA writes lane L to slot L, so lane L simply fetches the piece that belongs there -- conflict-free 16-byte
//     operand reads); three chunks in flight, one barrier per chunk;
//   * weights in MFMA operand order, an eight-step register ring (as bf16_trunk_layer_kernel); the four waves split the columns;
//   * the K loop is fully unrolled and EVERY memory instruction of it is issued in a fixed order (inline assembly), so the `vmcnt` immediates
//     are compile-time counts: 15 operations behind a weight load when it is used, 40 behind a chunk's DMA when it is read;
//   * epilogue = the plain trunk layer's (statistics on the fp32 accumulators, affine, bf16 store); the conv bias cancels under the norm.
// The eight channel tiles of a sample are mapped onto ONE XCD (workgroup id & 7 = XCD), so a sample's 1.3 MB are fetched into one L2.
constexpr int kC2K = 5120, kC2Steps = kC2K / 16, kC2Chunks = kC2K / 128, kC2StageB = 128 * 256;
template <int N> __device__ __forceinline__ void c2_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

__global__ void __launch_bounds__(256) bf16_c2d1d_kernel(const Bf16TrunkArgs a, int B)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int lin = blockIdx.x, slot = lin & 7, q = lin >> 3;
    const int tile = q & 7, n = (q >> 3) * 8 + slot;
    if (n >= B) return;
    float* red = reinterpret_cast<float*>(smem + 4 * kC2StageB);          // [4 waves][32 rows][2]
    // DMA sources: piece index i * 256 + tid of a chunk = (row i * 16 + (tid >> 4), slot tid & 15); the lane fetches logical piece slot ^ (row & 15)
    const bf16_t* xb = a.x + (long long)n * a.x_sn;
    const int prow = tid >> 4, pp = (tid & 15) ^ (prow & 15);
    const bf16_t* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { int r = i * 16 + prow; r = r < a.W ? r : a.W - 1; src[i] = xb + (long long)r * kC2K + pp * 8; }
    auto dma = [&](int chunk, int stage) __attribute__((always_inline)) {
        unsigned char* dst = smem + stage * kC2StageB + wave * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + chunk * 128),
                                             (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, 0, 0);
    };
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + (long long)tile * kC2Steps * 64 + lane;
    u32x4 aq[8];
    auto a_load = [&](int s, u32x4& dstv) __attribute__((always_inline)) {
        const u32x4* p = wp + (long long)s * 64;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dstv) : "v"(p) : "memory");
    };
    dma(0, 0); dma(1, 1); dma(2, 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) a_load(u, aq[u]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int px = wave * 32 + l31;
    const unsigned xs0 = lds_addr(smem) + (unsigned)px * 256u;
    const unsigned sw = (unsigned)(px & 15);
    u32x4 bv[2];
#pragma unroll
    for (int c = 0; c < kC2Chunks; ++c) {
        // chunk c has landed: operations issued behind its DMA = 24 / 32 / 40 / 40 / ... (fixed issue order: see the header)
        if (c == 0) c2_wait_vm<24>(); else if (c == 1) c2_wait_vm<32>(); else c2_wait_vm<40>();
        __builtin_amdgcn_s_barrier();                                 // ... for every wave's part; the stage chunk c - 1 used is free
        dma(c + 3 < kC2Chunks ? c + 3 : kC2Chunks - 1, (c + 3) & 3);  // (past the end: a harmless re-fetch into the free stage keeps the counts uniform)
        const unsigned cb = xs0 + (unsigned)((c & 3) * kC2StageB);
        asm volatile("ds_read_b128 %0, %1" : "=&v"(bv[0]) : "v"(cb + (((unsigned)half ^ sw) << 4)));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < 7) asm volatile("ds_read_b128 %0, %1" : "=&v"(bv[(u + 1) & 1]) : "v"(cb + (((unsigned)(2 * (u + 1) + half) ^ sw) << 4)));
            asm volatile("s_waitcnt vmcnt(15)" : "+v"(aq[u]));       // the weight operand of this step (requested eight steps ago)
            const u32x4 av = aq[u];
            const int sn = c * 8 + u + 8;
            a_load(sn < kC2Steps ? sn : kC2Steps - 1, aq[u]);
            if (u < 7) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bv[u & 1])); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[u & 1]));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv[u & 1]), acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- statistics over the valid columns, affine, store (as bf16_trunk_layer_kernel<false, .>)
    float s1[16], s2[16];
    const bool ok = px < a.W;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float v = ok ? acc[r] : 0.f; s1[r] = v; s2[r] = v * v; }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[((wave * 32 + row) << 1)] = s1[r]; red[((wave * 32 + row) << 1) + 1] = s2[r];
        }
    }
    __syncthreads();
    const float invW = 1.0f / (float)a.W;
    if (ok) {
        bf16_t* yp = a.y + (long long)n * a.y_sn + (long long)px * a.C + tile * 32;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * qd + j, row = j + 8 * qd + 4 * half;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) { t1 += red[((w4 * 32 + row) << 1)]; t2 += red[((w4 * 32 + row) << 1) + 1]; }
                const float mean = t1 * invW;
                float var = t2 * invW - mean * mean;
                if (var < 0.f) var = 0.f;
                const float sc = a.g0[tile * 32 + row] / sqrtf(var + a.eps);
                o[j] = (acc[r] - mean) * sc + a.b0[tile * 32 + row];
            }
            uint2 pk2; pk2.x = pack2bf(o[0], o[1]); pk2.y = pack2bf(o[2], o[3]);
            *reinterpret_cast<uint2*>(yp + 8 * qd + 4 * half) = pk2;
        }
    }
}

// conv2dto1d's weight in operand order: dst[((tile * 320 + s) * 64 + lane) * 8 + j] = W[32 tile + (lane & 31)][source channel of memory channel
// kf = 16 s + 8 (lane >> 5) + j], memory channel kf = h * 256 + c <-> source channel c * 20 + h (model.py:249-251 as the forward lays y3 out)
__global__ void __launch_bounds__(256) bf16_c2d1d_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst)
{
    const long long total = 8LL * kC2Steps * 512;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long ts = idx >> 9;
        const int s_ = (int)(ts % kC2Steps), tile = (int)(ts / kC2Steps);
        const int co = tile * 32 + (lane & 31), kf = 16 * s_ + 8 * (lane >> 5) + j;
        const int ci = (kf & 255) * 20 + (kf >> 8);
        dst[idx] = f2bf(w[(long long)co * kC2K + ci]);
    }
}

__global__ void __launch_bounds__(256) bf16_pack_kernel(const Bf16PackArgs a)
{
    const int ncc = a.Cin >> 5;
    const long long total = (long long)a.Cout_pad * a.KH * ncc * a.KW * 32;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx & 31);
        long long t = idx >> 5;
        const int kw = (int)(t % a.KW); t /= a.KW;
        const int cc = (int)(t % ncc); t /= ncc;
        const int kh = (int)(t % a.KH);
        int row = (int)(t / a.KH);
        const int k = cc * 32 + j;                   // packed input-channel index
        if (a.glu_interleave) {                      // packed row -> logical row of the [value | gate] concatenation
            const int blk = row >> 6, r = row & 63;
            row = (r < 32) ? (blk * 32 + r) : (a.Cout_pad / 2 + blk * 32 + (r - 32));
        }
        float v = 0.f;
        if (a.kind == BF16_PACK_KW_OUT) {
            if (row < a.KW_src && k < a.Cin_src) v = a.w[0][((long long)k * a.KH + kh) * a.KW_src + row];            // [0][ci][kh][kw = row]
        } else {
            int br = 0, co = row;
            if (a.kind == BF16_PACK_HC_OUT) { if (row < a.Cout_src) co = (row & 255) * 20 + (row >> 8); }               // row = h*256 + c  <- c*20 + h
            if (a.nbr == 2) { const int halfc = a.glu_interleave ? a.Cout_pad / 2 : a.Cout_src; br = co >= halfc ? 1 : 0; co -= br * halfc; }
            if (co < a.Cout_src) {
                if (a.kind == BF16_PACK_FOLD_KW) {
                    const int skw = k / a.Cin_src, ci = k - skw * a.Cin_src;
                    if (skw < a.KW_src) v = a.w[br][(((long long)co * a.Cin_src + ci) * a.KH + kh) * a.KW_src + skw];
                } else if (a.kind == BF16_PACK_HC_IN) {
                    const int kf = kw * a.Cin + k;               // (packed KW > 1: the 1 x 1 layer's channels in KW groups of Cin, see infer_bf16.hip)
                    if (kf < a.Cin_src) { const int ci = (kf & 255) * 20 + (kf >> 8); v = a.w[br][(long long)co * a.Cin_src + ci]; }
                } else {
                    // packed KW = G * KW_src: G channel groups of Cin as extra positions of a stride-G window (infer_bf16.hip, the 1-D trunk):
                    // window position kw = kw_src * G + g holds source channel g * Cin + k
                    int ks = k, kws = kw;
                    if (a.KW > a.KW_src) { const int G = a.KW / a.KW_src; kws = kw / G; ks = (kw % G) * a.Cin + k; }
                    if (ks < a.Cin_src) v = a.w[br][(((long long)co * a.Cin_src + ks) * a.KH + kh) * a.KW_src + kws];
                }
            }
        }
        a.dst[idx] = f2bf(v);
    }
}

// kind PLAIN: dst = [src | src2] (src2 optional, n_src elements each); HC_OUT: dst[h*256 + c] = src[c*20 + h];
// kind -1 (GLU interleave): 64-element blocks = [32 of src | 32 of src2]
__global__ void __launch_bounds__(256) bf16_vec_kernel(const float* __restrict__ src, const float* __restrict__ src2, float* __restrict__ dst,
                                                       int n_src, int n_dst, int kind)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_dst) return;
    float v = 0.f;
    if (kind == -1) {
        const int blk = i >> 6, r = i & 63, c = blk * 32 + (r & 31);
        if (c < n_src) v = (r < 32) ? src[c] : src2[c];
    } else if (kind == BF16_PACK_HC_OUT) {
        if (i < n_src) v = src[(i & 255) * 20 + (i >> 8)];
    } else {
        if (i < n_src) v = src[i];
        else if (src2 && i < 2 * n_src) v = src2[i - n_src];
    }
    dst[i] = v;
}

static unsigned ew_blocks(long long total)
{
    long long b = cdiv_ll(total, 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

int mcvc_bf16_prep_launch(const float* x, const float* mask, bf16_t* xin, int B, int H, int W, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, (double)B * H * W * (8.0 + 64.0));
    hipLaunchKernelGGL(bf16_prep_kernel, dim3(ew_blocks((long long)B * H * W * 4)), dim3(256), 0, s, x, mask, xin, B, H, W);
    return (int)hipGetLastError();
}

int mcvc_bf16_conv1_fused_launch(const float* x, const float* mask, const bf16_t* w, const float* bias, bf16_t* y, int B, int H, int W, hipStream_t s)
{
    constexpr int SH = 20;
    const int strips = cdiv_i(W, kC1W), segs = cdiv_i(H, SH);
    const double px = (double)B * H * W;
    TraceScope ts(K_CONV_L, s, 2.0 * px * 256 * 160, px * (8.0 + 2.0 * 128) + 2.0 * 256 * 160);
    hipLaunchKernelGGL(bf16_conv1_fused_kernel<SH>, dim3((unsigned)(B * segs * strips)), dim3(256), 0, s, x, mask, w, bias, y, H, W, segs);
    return (int)hipGetLastError();
}

int mcvc_bf16_last_launch(const bf16_t* z, const float* bias, float* out, int B, int H, int W, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, (double)B * H * W * (64.0 + 4.0));
    hipLaunchKernelGGL(bf16_last_kernel, dim3(ew_blocks((long long)B * H * W)), dim3(256), 0, s, z, bias, out, B, H, W);
    return (int)hipGetLastError();
}

int mcvc_bf16_last_fused_launch(const bf16_t* x, const bf16_t* w, const float* bias, float* out, int B, int H, int W, hipStream_t s)
{
    constexpr int SH = 10;
    const int strips = cdiv_i(W, kLastOut), segs = cdiv_i(H, SH);
    const double px = (double)B * H * W;
    TraceScope ts(K_CONV_L, s, 2.0 * px * 16 * 640, px * (256.0 + 4.0) + 2.0 * 32 * 640);
    hipLaunchKernelGGL(bf16_last_fused_kernel<SH>, dim3((unsigned)(B * segs * strips)), dim3(256), 0, s, x, w, bias, out, H, W, segs);
    return (int)hipGetLastError();
}

// ---- fused residual-block layer (bf16_trunk_layer_kernel): applies when a sample row fits one tile width and the channel counts fit the tiles
bool mcvc_bf16_trunk_layer_applies(int W, int Cin, int C) { return W >= 1 && W <= kTrunkMaxW && (Cin == 256 || Cin == 512) && (C % 32) == 0; }
long long mcvc_bf16_trunk_pack_elems(int Cin, int C, int glu) { return (long long)(glu ? 2 * C : C) * 3 * Cin; }

int mcvc_bf16_trunk_pack_launch(const float* w0, const float* w1, bf16_t* dst, int Cin, int C, int glu, hipStream_t s)
{
    const int tiles = (glu ? 2 * C : C) / 32;
    const long long total = (long long)tiles * (3 * Cin / 16) * 512;
    hipLaunchKernelGGL(bf16_trunk_pack_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, w0, w1, dst, Cin, tiles, glu);
    return (int)hipGetLastError();
}

int mcvc_bf16_trunk_layer_launch(const bf16_t* x, long long x_sn, const bf16_t* w, const float* g0, const float* b0, const float* g1, const float* b1,
                                 const bf16_t* res, bf16_t* y, long long y_sn, int B, int W, int Cin, int C, int glu, float eps, hipStream_t s)
{
    if (!mcvc_bf16_trunk_layer_applies(W, Cin, C)) return MCVC_ERR_INVALID;
    Bf16TrunkArgs a{};
    a.x = x; a.x_sn = x_sn; a.w = w; a.g0 = g0; a.b0 = b0; a.g1 = g1; a.b1 = b1; a.res = res; a.y = y; a.y_sn = y_sn;
    a.W = W; a.Cin = Cin; a.C = C; a.eps = eps;
    const size_t lds = (size_t)(kTrunkMaxW + 2) * (Cin * 2 + 16) + 2 * 4 * 32 * 2 * sizeof(float);
    const double rows = glu ? 2.0 * C : (double)C;
    TraceScope ts(K_CONV_L, s, 2.0 * B * W * rows * 3 * Cin, 2.0 * ((double)B * W * Cin * (C / 32) + (double)B * W * C * (res ? 2 : 1) + rows * 3 * Cin * B));
    static bool done = false;
    if (!done) {
        hipError_t e = hipSuccess;
#define MCVC_TRUNK_ATTR(G, CI) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_trunk_layer_kernel<G, CI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MCVC_TRUNK_ATTR(true, 256) MCVC_TRUNK_ATTR(true, 512) MCVC_TRUNK_ATTR(false, 256) MCVC_TRUNK_ATTR(false, 512)
#undef MCVC_TRUNK_ATTR
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    const dim3 grid((unsigned)(C / 32), (unsigned)B);
    if (glu && Cin == 256) hipLaunchKernelGGL((bf16_trunk_layer_kernel<true, 256>), grid, dim3(256), lds, s, a);
    else if (glu) hipLaunchKernelGGL((bf16_trunk_layer_kernel<true, 512>), grid, dim3(256), lds, s, a);
    else if (Cin == 256) hipLaunchKernelGGL((bf16_trunk_layer_kernel<false, 256>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((bf16_trunk_layer_kernel<false, 512>), grid, dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

// ---- conv2dto1d + InstanceNorm (bf16_c2d1d_kernel): a sample row of at most 128 columns
bool mcvc_bf16_c2d1d_applies(int W) { return W >= 1 && W <= kTrunkMaxW; }
long long mcvc_bf16_c2d1d_pack_elems(void) { return 256LL * kC2K; }
int mcvc_bf16_c2d1d_pack_launch(const float* w, bf16_t* dst, hipStream_t s)
{
    hipLaunchKernelGGL(bf16_c2d1d_pack_kernel, dim3(ew_blocks(8LL * kC2Steps * 512)), dim3(256), 0, s, w, dst);
    return (int)hipGetLastError();
}
int mcvc_bf16_c2d1d_launch(const bf16_t* x, long long x_sn, const bf16_t* w, const float* gamma, const float* beta, bf16_t* y, long long y_sn,
                           int B, int W, float eps, hipStream_t s)
{
    if (!mcvc_bf16_c2d1d_applies(W)) return MCVC_ERR_INVALID;
    Bf16TrunkArgs a{};
    a.x = x; a.x_sn = x_sn; a.w = w; a.g0 = gamma; a.b0 = beta; a.y = y; a.y_sn = y_sn; a.W = W; a.Cin = kC2K; a.C = 256; a.eps = eps;
    const size_t lds = 4 * (size_t)kC2StageB + 4 * 32 * 2 * sizeof(float);
    TraceScope ts(K_CONV_L, s, 2.0 * B * W * 256.0 * kC2K, 2.0 * ((double)B * W * kC2K * 8 + (double)B * W * 256 + 256.0 * kC2K * B));
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_c2d1d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    const int groups = cdiv_i(B, 8);
    hipLaunchKernelGGL(bf16_c2d1d_kernel, dim3((unsigned)(groups * 64)), dim3(256), lds, s, a, B);
    return (int)hipGetLastError();
}

int mcvc_bf16_pack_launch(const Bf16PackArgs& a, hipStream_t s)
{
    if ((a.Cin & 31) || a.nbr < 1 || a.nbr > 2) return MCVC_ERR_INVALID;
    const long long total = (long long)a.Cout_pad * a.KH * (a.Cin >> 5) * a.KW * 32;
    hipLaunchKernelGGL(bf16_pack_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int mcvc_bf16_vec_launch(const float* src, const float* src2, float* dst, int n_src, int n_dst, int kind, hipStream_t s)
{
    hipLaunchKernelGGL(bf16_vec_kernel, dim3((unsigned)cdiv_i(n_dst, 256)), dim3(256), 0, s, src, src2, dst, n_src, n_dst, kind);
    return (int)hipGetLastError();
}
