// Direct (im2col-free) convolution kernels for gfx950 on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
//  D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] for accumulator register r).
//
// Replaces the reference's nn.Conv2d / nn.Conv1d call sites (mask_cyclegan_vc/model.py:47-69, 86-99,
// 116-126, 142-146, 183-187, 227-231, 207-211, 290-327) -- forward, data-gradient (same kernel on dY with
// re-packed weights, see pack_kernels.hip) and weight-gradient.
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include <stdlib.h>
#include <stdio.h>

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#ifndef MCVC_CONV_MINW
#define MCVC_CONV_MINW 2
#endif

// =================================================================================================
// forward / data-gradient kernel
// =================================================================================================
// Block = 4 waves arranged BMW x BNW; each wave owns WM x WN accumulators of 32(co) x 32(pixel).
// LDS per channel chunk:  Xs[cic][PH][PWp]  (input patch incl. halo, zero padded)
//                         Ws[cic*KH*KW][COT] (K-major weight slice, straight copy of the packed rows)
// The two k-slots of the 32x32x2 MFMA are the even / odd channel of a channel pair, so both
// operand addresses are  lane_const + uniform_tap_offset  -> one v_add per ds_read_b32.
// Stride-2 patches are stored column-de-interleaved (even cols then odd cols) so the 32 lanes of a
// pixel row read consecutive banks; PWp is chosen on the host so that the rows of one 32-pixel
// sub-tile start in disjoint bank groups.
// Software pipeline, ONE barrier per channel chunk, everything double-buffered in LDS:
//   * the weight slice of chunk c+1 streams HBM/L2 -> LDS by direct DMA (global_load_lds, 16 B/lane, no
//     VGPRs); rows past the end of the packed matrix are redirected to its all-zero pad row;
//   * the (small) input patch of chunk c+1 is fetched into registers before the MFMA loop of chunk c and
//     written to the other LDS buffer after it (zero fill / stride-2 de-interleave happen on that write);
//   * __syncthreads() drains the DMA (hipcc emits vmcnt(0) in front of the barrier) and flips buffers.
constexpr int kMaxPR = 8;    // patch rows per half-wave            (cic*PH <= 64)
constexpr int kMaxPC = 3;    // 32-column groups per patch row      (PW <= 96)

// XCD-aware block remap (MI355X: 8 XCDs, private 4 MiB L2 each; hardware hands consecutive workgroup ids to
// consecutive XCDs).  Returns a logical id such that each XCD owns one contiguous range of logical ids, so
// workgroups that share an operand slice (same weights / same dY rows) hit the same L2.  Bijective for any total.
__device__ __forceinline__ int xcd_logical_id(int linear, int total)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = linear & 7, k = linear >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ float g_wgrad_zero[64];      // zero-initialised: DMA source for out-of-image elements

__device__ __forceinline__ void glds4(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

__device__ __forceinline__ void glds16(const float* g, float* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// KW is a template parameter: the kw taps of a kernel row are fully unrolled and their LDS offsets are
// ds_read immediates.  (With runtime kh/kw/stride arithmetic each tap cost ~30 SALU instructions; the scalar unit
// is shared by the CU's 8 resident waves, which made the loop SALU-bound at ~2x the MFMA time.)
template <int WM, int WN, int BMW, int BNW, int KW, bool S2>
__global__ void __launch_bounds__(256, MCVC_CONV_MINW) conv_direct_kernel(const Twin<ConvArgs> tw)
{
    const ConvArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int COT = 32 * WM * BMW;
    constexpr int NSUB = WN * BNW;
    constexpr int V = COT / 4;
    constexpr int RPI = 256 / V;                     // packed-weight rows covered by one 256-thread DMA sweep
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int hw = tid >> 5;                         // half-wave id 0..7
    const int wm_id = wave / BNW, wn_id = wave % BNW;
    const int KHKW = a.KH * KW;

    // logical order: pixel tile fastest, then image, then K-split, then co-tile -> the workgroups of one XCD
    // share (co-tile, split), i.e. the same weight slice, which then stays resident in that XCD's L2
    int lid = xcd_logical_id((int)blockIdx.x, (int)gridDim.x);
    const int tile = lid % a.tiles_total; lid /= a.tiles_total;
    const int n = lid % a.nb; lid /= a.nb;
    const int split = lid % a.nsplit;
    const int co0 = (lid / a.nsplit) * COT;
    const int tile_x = tile % a.tiles_w;
    const int tile_y = tile / a.tiles_w;

    const int tow = 1 << a.tow_log2;
    const int rps = 32 >> a.tow_log2;              // output rows per 32-pixel sub-tile
    const int toh = NSUB * rps;
    const int oh0 = tile_y * toh, ow0 = tile_x * tow;
    const int ih0 = oh0 * a.stride - a.pad_h;
    const int iw0 = ow0 * a.stride - a.pad_w;
    constexpr bool s2 = S2;                          // stride 2 (de-interleaved patch columns) is a compile-time variant

    // LDS map (float offsets into smem): [X buf0][X buf1][W buf0][W buf1].  Buffers are addressed as
    // smem[offset] -- never through a selected pointer, which would decay to a flat pointer and turn every
    // operand read into flat_load_dword + vmcnt(0).
    const int ws_floats = a.cic * KHKW * COT;
    const int wbase = 2 * a.xs_floats;

    const int r_j = l31 >> a.tow_log2, c_j = l31 & (tow - 1);
    int b_lane[WN];
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int sub = wn_id * WN + wn;
        b_lane[wn] = half * a.plane + (sub * rps + r_j) * a.stride * a.PWp + c_j;
    }
    const int a_lane = half * KHKW * COT + wm_id * (WM * 32) + l31;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ch_begin = split * a.chunks_per_split;
    int ch_end = ch_begin + a.chunks_per_split;
    if (ch_end > a.nchunks) ch_end = a.nchunks;

    const float* xn = a.x + (long long)n * a.x_sb;
    const int prow_n = a.cic * a.PH;               // patch rows per chunk
    const int prow_it = (prow_n + 7) >> 3;         // row sweeps actually needed (<= kMaxPR)
    const int pcol_it = (a.PW + 31) >> 5;          // column groups actually needed (<= kMaxPC)
    const int wrows = a.cic * KHKW;                // packed-weight rows per chunk
    // DMA lane constants: thread t moves float4 #(t % V) of row (t / V) + i*RPI on sweep i
    const int d_row0 = tid / V;
    int d_co = co0 + (tid % V) * 4;
    if (d_co > a.w_cout - 4) d_co = a.w_cout - 4;  // column tile past the row end: finite neighbours, outputs discarded
    const float* d_src0 = a.w + (long long)n * a.w_nstride + d_co;

    float preg[kMaxPR][kMaxPC];

    auto dma_weights = [&](int ch, int wdst_off) {
        const long long grow0 = (long long)ch * wrows;
        int row = d_row0;
        int lds = wdst_off + (wave * 64) * 4;                       // wave-uniform LDS base (float index)
        for (; row < wrows; row += RPI, lds += 1024) {
            long long grow = grow0 + row;
            if (grow > a.w_rows) grow = a.w_rows;                    // all-zero pad row
            glds16(d_src0 + grow * a.w_cout, smem + lds);
        }
    };
    auto fetch_patch = [&](int ch) {
        const int c0 = ch * a.cic;
#pragma unroll
        for (int k = 0; k < kMaxPR; ++k) {
            if (k < prow_it) {
                const int row = hw + 8 * k;
                const int ci = row / a.PH;
                const int r = row - ci * a.PH;
                const int ih = ih0 + r;
                const int cg = c0 + ci;
                const bool rok = (row < prow_n) && (cg < a.Cin) && (ih >= 0) && (ih < a.H);
                const float* src = xn + (long long)cg * a.x_sc + (long long)ih * a.x_sh + iw0 + l31;
#pragma unroll
                for (int j = 0; j < kMaxPC; ++j) {
                    if (j < pcol_it) {
                        const int iw = iw0 + l31 + 32 * j;
                        float v = 0.f;
                        if (rok && (l31 + 32 * j) < a.PW && iw >= 0 && iw < a.W) v = src[32 * j];
                        preg[k][j] = v;
                    }
                }
            }
        }
    };
    auto commit_patch = [&](int xdst_off) {
#pragma unroll
        for (int k = 0; k < kMaxPR; ++k) {
            if (k < prow_it) {
                const int row = hw + 8 * k;
                if (row < prow_n) {
                    const int ci = row / a.PH;
                    const int r = row - ci * a.PH;
                    const int dst = xdst_off + ci * a.plane + r * a.PWp;
#pragma unroll
                    for (int j = 0; j < kMaxPC; ++j) {
                        if (j < pcol_it) {
                            const int c = l31 + 32 * j;
                            if (c < a.PW) smem[dst + (s2 ? ((c & 1) * a.PWh + (c >> 1)) : c)] = preg[k][j];
                        }
                    }
                }
            }
        }
    };

    // GEMM mode (1x1, unit stride, no padding, image width == tile width): the pixel tile of a channel is `plane` contiguous
    // floats in memory, so the patch is streamed by the same 16-byte LDS-DMA as the weights -- no register prefetch, and the
    // channel chunk is no longer limited by the prefetch registers
    auto dma_patch = [&](int ch, int xdst_off) {
        const float* src0 = xn + (long long)(ch * a.cic) * a.x_sc + (long long)oh0 * a.x_sh;
        const int f4_per_ch = a.plane >> 2;
        const int total = a.cic * f4_per_ch;                 // float4s in the chunk; 64 per wave instruction
        for (int i0 = wave * 64; i0 < total; i0 += 256) {
            const int i = i0 + lane;
            const int ci = i / f4_per_ch, q = i - ci * f4_per_ch;
            if (i < total) glds16(src0 + (long long)ci * a.x_sc + 4 * q, smem + xdst_off + i0 * 4);
        }
    };
    if (ch_begin < ch_end) {
        dma_weights(ch_begin, wbase);
        if (a.gemm) dma_patch(ch_begin, 0);
        else { fetch_patch(ch_begin); commit_patch(0); }
    }
    __syncthreads();
    const int npairs = a.cic >> 1;
    const int odd_off = s2 ? a.PWh : 0;            // stride 2: odd patch columns live PWh floats after the even ones
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int cur = (ch - ch_begin) & 1;
        const bool more = (ch + 1 < ch_end);
        if (more) {
            dma_weights(ch + 1, wbase + (cur ^ 1) * ws_floats);
            if (a.gemm) dma_patch(ch + 1, (cur ^ 1) * a.xs_floats); else fetch_patch(ch + 1);
        }
        // ---- MFMA main loop: (channel pair, kh) rows at run time, the KW taps of a row unrolled with immediate
        // offsets.  Operand reads are software-pipelined one tap ahead (the reads of tap t+1 -- or of the next row's
        // first tap -- are issued before the MFMAs of tap t), so the LDS latency hides behind the matrix pipe instead of
        // sitting between every pair of MFMAs.
        if (KW == 1 && a.KH == 1) {
            // 1x1 (GEMM-shaped: the Winograd products, the generic trunk): one tap per channel pair -- a plain K loop with
            // fixed operand strides, unrolled so that several pairs' operand reads are in flight
            {
                const int a0 = wbase + cur * ws_floats + a_lane;
                const int x0 = cur * a.xs_floats;
#pragma unroll 4
                for (int cp = 0; cp < npairs; ++cp) {
                    float av[WM], bv[WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i) av[i] = smem[a0 + cp * 2 * COT + i * 32];
#pragma unroll
                    for (int j = 0; j < WN; ++j) bv[j] = smem[x0 + cp * 2 * a.plane + b_lane[j]];
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) acc[i][j] = MFMA32(av[i], bv[j], acc[i][j]);
                }
            }
        } else {
            int a_off = wbase + cur * ws_floats + a_lane;
            int x_row = cur * a.xs_floats;
            const int nrows = npairs * a.KH;
            float av[WM], bv[WN];
            auto load_tap = [&](int aoff, int xrow, int kw, float (&ra)[WM], float (&rb)[WN]) {
#pragma unroll
                for (int i = 0; i < WM; ++i) ra[i] = smem[aoff + kw * COT + i * 32];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    // stride 1: column kw.  stride 2 (de-interleaved): even kw -> kw/2, odd kw -> PWh + kw/2
                    const int xo = xrow + b_lane[j];
                    rb[j] = s2 ? ((kw & 1) ? smem[xo + odd_off + (kw >> 1)] : smem[xo + (kw >> 1)]) : smem[xo + kw];
                }
            };
            load_tap(a_off, x_row, 0, av, bv);
            int kh = 0;
            for (int row = 0; row < nrows; ++row) {
                int a_next = a_off + KW * COT, x_next = x_row + a.PWp;
                if (++kh == a.KH) {                 // next channel pair: skip the odd channel's weight rows / plane
                    kh = 0;
                    a_next += KHKW * COT;
                    x_next += 2 * a.plane - a.KH * a.PWp;
                }
                if (row + 1 == nrows) { a_next = a_off; x_next = x_row; }      // nothing follows: harmless re-read
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    float nav[WM], nbv[WN];
                    if (kw + 1 < KW) load_tap(a_off, x_row, kw + 1, nav, nbv);
                    else load_tap(a_next, x_next, 0, nav, nbv);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) acc[i][j] = MFMA32(av[i], bv[j], acc[i][j]);
                    // pin the order "reads of the next tap, then this tap's MFMAs": left alone, the scheduler sinks the
                    // reads below the MFMAs to shorten live ranges and re-exposes the LDS latency
                    __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, WM * WN, 0);
#pragma unroll
                    for (int i = 0; i < WM; ++i) av[i] = nav[i];
#pragma unroll
                    for (int j = 0; j < WN; ++j) bv[j] = nbv[j];
                }
                a_off = a_next; x_row = x_next;
            }
        }
        if (more && !a.gemm) commit_patch((cur ^ 1) * a.xs_floats);
        __syncthreads();
    }

    // ---- epilogue: bias, (shuffled) store / slab store / accumulate
    float* ybase = (split == 0 || a.out_mode == CONV_OUT_ACCUM) ? a.y : (a.y_slabs + (long long)(split - 1) * a.slab_stride);
    ybase += (long long)n * a.y_sb;
    const bool add_bias = (a.bias != nullptr) && (split == 0);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int sub = wn_id * WN + j;
        const int oh = oh0 + sub * rps + r_j;
        const int ow = ow0 + c_j;
        const bool pok = (oh < a.OH) && (ow < a.OW);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm_id * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pok && co < a.Cout) {
                    float v = acc[i][j][r];
                    if (add_bias) v += a.bias[co];
                    long long off;
                    if (a.shuffle) {
                        const int yh = 2 * oh + ((co >> 1) & 1), yw = 2 * ow + (co & 1);
                        if (yh >= a.YH || yw >= a.YW) continue;
                        off = (long long)(co >> 2) * a.y_sc + (long long)yh * a.y_sh + yw;
                    }
                    else
                        off = (long long)co * a.y_sc + (long long)oh * a.y_sh + (long long)ow * a.y_sw;
                    if (a.out_mode == CONV_OUT_ACCUM) {
                        if (a.nsplit > 1) unsafeAtomicAdd(ybase + off, v);   // global_atomic_add_f32 (no CAS loop)
                        else ybase[off] += v;
                    } else {
                        ybase[off] = v;
                    }
                }
            }
        }
    }
}

// ---- host planner --------------------------------------------------------------------------------
namespace {

// tuning knobs (mcvc_common.h mcvc_knob: constants in the product, MCVC_<NAME> in an experiments build)
static int env_int(const char* name, int dflt) { return mcvc_knob(name, dflt); }
// planner knobs (tools/conv_tune.py): read once, or on every call when MCVC_CONV_TUNE=1 was set at load time
static int conv_knob(const char* name, int dflt)
{
    static const int tune = env_int("MCVC_CONV_TUNE", 0);
    if (tune) return env_int(name, dflt);
    return dflt;
}
static int conv_lds_budget_floats() { static const int v = env_int("MCVC_CONV_LDS_KB", 78) * 256; return conv_knob("MCVC_CONV_LDS_KB", v / 256) * 256; }

enum ConvCfg { CFG_L = 0, CFG_M, CFG_N, CFG_T, CFG_S, CFG_S2, CFG_COUNT };
struct CfgDesc { int cot, npix, kind; };
static const CfgDesc kCfg[CFG_COUNT] = {
    {128, 128, K_CONV_L},   // L: 2x2 waves of 2x2 accumulators
    {128, 64, K_CONV_M},    // M: 2x2 waves of 2x1
    {64, 128, K_CONV_N},    // N: 2x2 waves of 1x2
    {256, 32, K_CONV_T},    // T: 4x1 waves of 2x1   (1-D trunk: few pixels, many channels)
    {32, 256, K_CONV_S},    // S: 1x4 waves of 1x2   (Cout <= 32)
    {32, 128, K_CONV_S2},   // S2: 1x4 waves of 1x1  (Cout <= 32, small images)
};

constexpr int kLdsBudgetFloats = 19 * 1024 + 512;   // 78 KiB per workgroup -> 2 workgroups per CU (160 KiB LDS)

struct ConvPlan {
    ConvArgs a;
    dim3 grid;
    int cfg;
    size_t lds_bytes;
};

static int pick_tow_log2(int OW) { return OW > 16 ? 5 : (OW > 8 ? 4 : 3); }

static void patch_geometry(const ConvProblem& p, int npix, int tow_log2, int* PH, int* PW, int* PWp, int* PWh)
{
    const int tow = 1 << tow_log2, rps = 32 >> tow_log2, toh = (npix / 32) * rps;
    *PH = (toh - 1) * p.stride + p.KH;
    *PW = (tow - 1) * p.stride + p.KW;
    *PWh = (*PW + 1) / 2;
    int need = (p.stride == 2) ? 2 * (*PWh) : *PW;
    int pwp = need;
    if (rps > 1) {
        // rows of one sub-tile must start in disjoint bank groups: (stride*PWp) % 32 == tow (or 32-tow when tow==8)
        for (;; ++pwp) {
            const int m = (p.stride * pwp) % 32;
            if (m == tow || (tow == 8 && m == 24)) break;
        }
    }
    *PWp = pwp;
}

static bool make_plan(const ConvProblem& p, int NB, int allow_split, int force_nsplit, ConvPlan* out, int force_cfg = -1, int allow_gemm = 0)
{
    if (p.stride != 1 && p.stride != 2) return false;
    if (p.KW != 1 && p.KW != 2 && p.KW != 3 && p.KW != 5 && p.KW != 15) return false;
    if (p.stride == 2 && p.KW != 3 && p.KW != 5) return false;
    const int tow_log2 = pick_tow_log2(p.OW);
    const int tow = 1 << tow_log2, rps = 32 >> tow_log2;
    const int npix_img = p.OH * p.OW;
    // candidate tile configurations in order of preference; the first one whose LDS / register-prefetch
    // geometry fits is used
    // Rules distilled from the per-layer sweep of tools/conv_tune.py on gfx950 (profiles/r01_conv_tune.log):
    //  * <= 3x3-sized filters on small images (<= 320 px) with many output channels stream weights: the 256-channel
    //    x 32-pixel tile (T) wins by 10-45 %;
    //  * few output channels (<= 256): the 32-channel x 128-pixel tile (S2) -- many more workgroups, and as fast per
    //    MFMA as the big tiles because every wave keeps only one accumulator (more resident waves);
    //  * otherwise the 128x128 tile (L) when the image is large enough to tile without waste and split-K can still
    //    fill the chip, else 128x64 (M) under the same condition, else S2.
    int cand[4], ncand = 0;
    const int khkw_ = p.KH * p.KW;
    if (p.Cout <= 32) {
        if (npix_img <= 320 && khkw_ <= 9) cand[ncand++] = CFG_T;
        cand[ncand++] = CFG_S2; cand[ncand++] = CFG_S;
    } else if (khkw_ <= 9 && npix_img <= 320 && p.Cout >= 256) {
        cand[ncand++] = CFG_T; cand[ncand++] = CFG_M; cand[ncand++] = CFG_N;
    } else if (p.Cout <= 256) {
        cand[ncand++] = CFG_S2; cand[ncand++] = CFG_M; cand[ncand++] = CFG_N;
    } else {
        const int toh_l = (kCfg[CFG_L].npix / 32) * rps, toh_m = (kCfg[CFG_M].npix / 32) * rps;
        const long long blocks_l = (long long)cdiv_i(p.OW, tow) * cdiv_i(p.OH, toh_l) * cdiv_i(p.Cout, 128) * NB;
        const long long blocks_m = (long long)cdiv_i(p.OW, tow) * cdiv_i(p.OH, toh_m) * cdiv_i(p.Cout, 128) * NB;
        const int max_split = 16;          // (the same for the planning and the launching call: both must pick one tile)
        if (npix_img >= 1024 && blocks_l * max_split >= 512 && khkw_ <= 25) cand[ncand++] = CFG_L;
        else if (blocks_m * max_split < 512) cand[ncand++] = CFG_S2;
        cand[ncand++] = CFG_M; cand[ncand++] = CFG_N;
    }
    if (force_cfg >= 0 && force_cfg < CFG_COUNT) { cand[0] = force_cfg; ncand = 1; }
    if (conv_knob("MCVC_CONV_CFG", -1) >= 0) { cand[0] = conv_knob("MCVC_CONV_CFG", -1); ncand = 1; }
    const int khkw = p.KH * p.KW;
    const int cin_pad = round_up_i(p.Cin, 2);
    for (int t = 0; t < ncand; ++t) {
        ConvPlan pl;
        ConvArgs& a = pl.a;
        a = ConvArgs{};
        const int cfg = cand[t];
        const int cot = kCfg[cfg].cot, npix = kCfg[cfg].npix;
        const int toh = (npix / 32) * rps;
        patch_geometry(p, npix, tow_log2, &a.PH, &a.PW, &a.PWp, &a.PWh);
        a.plane = a.PH * a.PWp;
        if (a.PW > 32 * kMaxPC) continue;
        // both operands are double-buffered: 2 * (patch + weights) floats
        const int budget = conv_lds_budget_floats();
        auto fits = [&](int c) {
            return c <= cin_pad && 2 * (round_up_i(c * a.plane, 4) + c * khkw * cot) <= budget && c * khkw <= 256 &&
                   c * a.PH <= 8 * kMaxPR;
        };
        int cic = 2;
        while (fits(cic + 2)) cic += 2;
        // GEMM mode: both operands by DMA (see the kernel); the chunk must divide Cin (no partially valid chunk: the
        // un-fetched tail would be stale LDS, and 0 * NaN is NaN)
        a.gemm = 0;
        if (allow_gemm && khkw == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && p.W == tow && p.OW == tow && tow == 32 && (a.plane & 3) == 0 &&
            a.PWp == tow && (p.Cin & 1) == 0) {
            int best = 0;
            for (int c = 2; c <= p.Cin && c <= 128; c += 2)
                if (p.Cin % c == 0 && 2 * (c * a.plane + c * cot) <= budget) best = c;
            if (best >= 8) { a.gemm = 1; cic = best; }
        }
        if (2 * (round_up_i(cic * a.plane, 4) + cic * khkw * cot) > 40 * 1024) continue;   // would not fit 160 KiB
        if (!a.gemm && cic * a.PH > 8 * kMaxPR) continue;
        a.cic = cic;
        a.xs_floats = round_up_i(cic * a.plane, 4);
        a.nchunks = cdiv_i(cin_pad, cic);
        a.tow_log2 = tow_log2;
        a.tiles_w = cdiv_i(p.OW, tow);
        const int tiles_h = cdiv_i(p.OH, toh);
        const int cotiles = cdiv_i(p.Cout, cot);
        const long long blocks = (long long)a.tiles_w * tiles_h * cotiles * NB;
        int nsplit = 1;
        if (force_nsplit > 0) {
            // exact count requested: trailing splits may own no chunk and then only write zeros (+bias)
            a.nsplit = force_nsplit;
            a.chunks_per_split = cdiv_i(a.nchunks, force_nsplit);
        } else {
            if (allow_split && blocks < 256 && a.nchunks > 1) {
                nsplit = (int)(512 / blocks);             // one resident round: 2 workgroups per CU x 256 CUs
                if (nsplit > a.nchunks) nsplit = a.nchunks;
                if (nsplit > 16) nsplit = 16;             // every slab is re-read by the consumer kernel
                if (nsplit < 1) nsplit = 1;
            }
            if (allow_split && conv_knob("MCVC_CONV_NSPLIT", 0) > 0) {
                nsplit = conv_knob("MCVC_CONV_NSPLIT", 0);
                if (nsplit > a.nchunks) nsplit = a.nchunks;
            }
            a.chunks_per_split = cdiv_i(a.nchunks, nsplit);
            a.nsplit = cdiv_i(a.nchunks, a.chunks_per_split);
        }
        a.Cin = p.Cin; a.H = p.H; a.W = p.W;
        a.Cout = p.Cout; a.OH = p.OH; a.OW = p.OW;
        a.KH = p.KH; a.KW = p.KW; a.stride = p.stride; a.pad_h = p.pad_h; a.pad_w = p.pad_w;
        pl.cfg = cfg;
        a.tiles_total = a.tiles_w * tiles_h; a.nb = NB;
        pl.grid = dim3((unsigned)(a.tiles_total * cotiles * NB * a.nsplit));
        pl.lds_bytes = (size_t)2 * (a.xs_floats + cic * khkw * cot) * sizeof(float);
        if (conv_knob("MCVC_CONV_VERBOSE", 0))
            fprintf(stderr, "[conv plan] Cin=%d Cout=%d k=%dx%d s=%d NB=%d out=%dx%d | cfg=%d cic=%d chunks=%d nsplit=%d grid=%u lds=%zu\n",
                    p.Cin, p.Cout, p.KH, p.KW, p.stride, NB, p.OH, p.OW, cfg, cic, a.nchunks, a.nsplit, pl.grid.x, pl.lds_bytes);
        *out = pl;
        return true;
    }
    return false;
}

template <int WM, int WN, int BMW, int BNW, int KW, bool S2>
static hipError_t launch_cfg_kw(const ConvPlan& pl, hipStream_t s)
{
    auto kern = conv_direct_kernel<WM, WN, BMW, BNW, KW, S2>;
    static bool attr_done = false;             // once per instantiation (benign race: idempotent)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const ConvArgs& a = pl.a;
    const int nb = a.nb;
    const double px = (double)nb * a.OH * a.OW;
    TraceScope ts(kCfg[pl.cfg].kind, s, 2.0 * px * a.Cout * a.Cin * a.KH * a.KW,
                  4.0 * ((double)nb * a.Cin * a.H * a.W + (double)a.Cin * a.KH * a.KW * a.Cout + px * a.Cout * a.nsplit));
    mcvc_launch(kern, pl.grid, dim3(256), pl.lds_bytes, s, pl.a);
    return hipGetLastError();
}

template <int WM, int WN, int BMW, int BNW>
static hipError_t launch_cfg(const ConvPlan& pl, hipStream_t s)
{
    if (pl.a.stride == 2) {                     // the network's strided convs are 5x5 (generator) and 3x3 (discriminator)
        switch (pl.a.KW) {
            case 3: return launch_cfg_kw<WM, WN, BMW, BNW, 3, true>(pl, s);
            case 5: return launch_cfg_kw<WM, WN, BMW, BNW, 5, true>(pl, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (pl.a.KW) {
        case 1: return launch_cfg_kw<WM, WN, BMW, BNW, 1, false>(pl, s);
        case 2: return launch_cfg_kw<WM, WN, BMW, BNW, 2, false>(pl, s);
        case 3: return launch_cfg_kw<WM, WN, BMW, BNW, 3, false>(pl, s);
        case 5: return launch_cfg_kw<WM, WN, BMW, BNW, 5, false>(pl, s);
        case 15: return launch_cfg_kw<WM, WN, BMW, BNW, 15, false>(pl, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

int mcvc_conv_plan_nsplit(const ConvProblem& p, int NB, int allow_split)
{
    if (mcvc_fewout_applies(p)) return mcvc_fewout_plan_nsplit(p, NB, allow_split);
    ConvPlan pl;
    if (!make_plan(p, NB, allow_split, 0, &pl)) return -1;
    return pl.a.nsplit;
}

int mcvc_conv_launch(const ConvProblem& p, int NB, const ConvIO& io, const float* wpk, int w_rows, int w_cout,
                     const float* bias, hipStream_t s, int* nsplit_out)
{
    ConvPlan pl;
    if (io.nsplit < 1) return MCVC_ERR_INVALID;
    if (io.nsplit > 1 && !io.accumulate && io.slabs == nullptr) return MCVC_ERR_WORKSPACE;
    if (mcvc_fewout_applies(p) && !io.shuffle) {
        if (nsplit_out) *nsplit_out = io.nsplit;
        return mcvc_fewout_launch(p, NB, io, wpk, w_cout, bias, s);
    }
    const int gemm_ok = io.gemm_ok && io.x_sh == p.W && (io.x_sc & 3) == 0 && (io.x_sb & 3) == 0 && ((uintptr_t)io.x & 15) == 0;
    if (!make_plan(p, NB, 0, io.nsplit, &pl, io.tile_cfg - 1, gemm_ok)) return MCVC_ERR_INVALID;
    ConvArgs& a = pl.a;
    if ((w_cout & 3) != 0) return MCVC_ERR_INVALID;
    a.x = io.x; a.x_sb = io.x_sb; a.x_sc = io.x_sc; a.x_sh = io.x_sh;
    a.y = io.y; a.y_sb = io.y_sb; a.y_sc = io.y_sc; a.y_sh = io.y_sh; a.y_sw = io.y_sw;
    a.y_slabs = io.slabs; a.slab_stride = io.slab_stride;
    a.w = wpk; a.w_rows = w_rows; a.w_cout = w_cout; a.bias = bias; a.w_nstride = io.w_nstride;
    a.out_mode = io.accumulate ? CONV_OUT_ACCUM : CONV_OUT_SLAB;
    a.shuffle = io.shuffle;
    a.YH = io.YH > 0 ? io.YH : 2 * p.OH; a.YW = io.YW > 0 ? io.YW : 2 * p.OW;
    if (nsplit_out) *nsplit_out = a.nsplit;
    hipError_t e;
    switch (pl.cfg) {
        case CFG_L: e = launch_cfg<2, 2, 2, 2>(pl, s); break;
        case CFG_M: e = launch_cfg<2, 1, 2, 2>(pl, s); break;
        case CFG_N: e = launch_cfg<1, 2, 2, 2>(pl, s); break;
        case CFG_T: e = launch_cfg<2, 1, 4, 1>(pl, s); break;
        case CFG_S2: e = launch_cfg<1, 1, 1, 4>(pl, s); break;
        default:    e = launch_cfg<1, 2, 1, 4>(pl, s); break;
    }
    return (int)e;
}

// =================================================================================================
// weight-gradient kernel
// =================================================================================================
// Block = nwaves waves; wave t owns task (co group, kh, kw group) = MS x KWT accumulators of
// 32(co) x 32(ci).  Per pixel chunk the block stages dYs[cot][pitch_a] and Xs[32][PH][PWp] (odd
// pitches -> conflict-free lane strides) and every wave walks the chunk's pixel pairs.
// Write-out: accumulators are transposed through LDS 8 output channels at a time so that every
// output channel's [ci][kh][kw] run (contiguous in OIHW) is written with coalesced stores -- either
// `dw +=` directly (ksplit == 1, deterministic) or to this K-split's private slab, which
// wgrad_reduce_kernel then folds into dw (no atomics anywhere).
template <int MS, int KWT, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) conv_wgrad_kernel(const Twin<WgradArgs> tw)
{
    const WgradArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int nwaves_ = nthreads >> 6;

    const int tasks_per_group = a.KH * a.nkwg;
    const int cgrp = wave / tasks_per_group;
    const int trem = wave - cgrp * tasks_per_group;
    const int kh = trem / a.nkwg;
    const int kw0 = (trem - kh * a.nkwg) * KWT;
    const int ms_base = cgrp * MS * 32;
    const int KHKW = a.KH * a.KW;

    // logical order: ci tile fastest, then K-split, then co tile -> an XCD's workgroups share dY rows
    int lid = xcd_logical_id((int)blockIdx.x, (int)gridDim.x);
    const int ci0 = (lid % a.ci_tiles) * 32; lid /= a.ci_tiles;
    const int zsplit = lid % a.ksplit;
    const int co0 = (lid / a.ksplit) * a.cot;

    float* As = smem;                           // [cot][pitch_a]
    float* Xs = smem + a.cot * a.pitch_a;       // [nci][plane]

    int lane_off, lane_ci, lane_kw;
    bool lane_ok;
    if (a.lane_mode == 0) {
        lane_ci = l31; lane_kw = 0; lane_ok = (ci0 + l31) < a.Cin;
        lane_off = l31 * a.plane;
    } else {
        lane_ci = l31 / a.KW; lane_kw = l31 - lane_ci * a.KW; lane_ok = lane_ci < a.Cin;
        lane_off = lane_ok ? (lane_ci * a.plane + lane_kw) : 0;
    }
    lane_off += half * a.stride;                // odd pixel of the pair

    f32x16 acc[MS][KWT];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int t = 0; t < KWT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    const int tiles = a.tiles_h * a.tiles_w;
    const int items = a.NB * tiles;
    const int nci = (a.lane_mode == 0) ? 32 : a.Cin;
    const int ci_base = (a.lane_mode == 0) ? ci0 : 0;
    const int tw2 = a.tow >> 1;
    const int rows_a = a.cot * a.toh, rows_x = nci * a.PH;

    for (int item = zsplit; item < items; item += a.ksplit) {
        const int n = item / tiles;
        const int t = item - n * tiles;
        const int ty = t / a.tiles_w, tx = t - ty * a.tiles_w;
        const int oh0 = ty * a.toh, ow0 = tx * a.tow;
        const int ih0 = oh0 * a.stride - a.pad_h, iw0 = ow0 * a.stride - a.pad_w;
        __syncthreads();
        // Staging by LDS-DMA (global_load_lds, 4 B per lane, no VGPRs): a wave fills 64 consecutive LDS floats per
        // instruction, each lane fetching its own (bounds-checked) global element or a zero; all of a wave's requests
        // are in flight together and drained once by the vmcnt(0) in front of the barrier.  (The synchronous
        // load->ds_write loop this replaces was latency-bound: ~100 dependent round trips per wave and tile.)
        {
            // dY: per output channel one contiguous run of toh*tow floats (rows of the pixel tile)
            const float* dyn = a.dy + (long long)n * a.dy_sb;
            const int LA = a.toh * a.tow;
            const int chunks = (LA + 63) >> 6;
            const int tow_log2 = 31 - __builtin_clz((unsigned)a.tow);
            for (int job = wave; job < a.cot * chunks; job += nwaves_) {
                const int co = job / chunks, ch = job - co * chunks;
                const int idx = ch * 64 + lane;
                const int r = idx >> tow_log2, c = idx & (a.tow - 1);
                const int oh = oh0 + r, ow = ow0 + c, cg = co0 + co;
                const bool ok = (cg < a.Cout) && (oh < a.OH) && (ow < a.OW);
                const float* src = ok ? (dyn + (long long)cg * a.dy_sc + (long long)oh * a.dy_sh + ow) : (g_wgrad_zero + lane);
                if (idx < LA) glds4(src, As + co * a.pitch_a + ch * 64);
            }
        }
        {
            // X: per input channel one contiguous run of PH*PW floats (the haloed patch, PWp == PW)
            const float* xn = a.x + (long long)n * a.x_sb;
            const int LX = a.PH * a.PW;
            const int chunks = (LX + 63) >> 6;
            const float inv_pw = 1.0f / (float)a.PW;
            for (int job = wave; job < nci * chunks; job += nwaves_) {
                const int ci = job / chunks, ch = job - ci * chunks;
                const int idx = ch * 64 + lane;
                int r = (int)((float)idx * inv_pw);                     // idx < 2^16: exact after the fix-up below
                int c = idx - r * a.PW;
                if (c < 0) { c += a.PW; --r; } else if (c >= a.PW) { c -= a.PW; ++r; }
                const int ih = ih0 + r, iw = iw0 + c, cg = ci_base + ci;
                const bool ok = (cg < a.Cin) && (ih >= 0) && (ih < a.H) && (iw >= 0) && (iw < a.W);
                const float* src = ok ? (xn + (long long)cg * a.x_sc + (long long)ih * a.x_sh + iw) : (g_wgrad_zero + lane);
                if (idx < LX) glds4(src, Xs + ci * a.plane + ch * 64);
            }
        }
        __syncthreads();
        const float* ap = As + (ms_base + l31) * a.pitch_a + half;
        for (int r = 0; r < a.toh; ++r) {
            const float* xr = Xs + lane_off + (r * a.stride + kh) * a.PWp + ((a.lane_mode == 0) ? kw0 : 0);
            const float* ar = ap + r * a.tow;
            for (int cp = 0; cp < tw2; ++cp) {
                float av[MS], bv[KWT];
#pragma unroll
                for (int i = 0; i < MS; ++i) av[i] = ar[i * 32 * a.pitch_a + 2 * cp];
#pragma unroll
                for (int t2 = 0; t2 < KWT; ++t2) bv[t2] = xr[2 * cp * a.stride + t2];
#pragma unroll
                for (int i = 0; i < MS; ++i)
#pragma unroll
                    for (int t2 = 0; t2 < KWT; ++t2) acc[i][t2] = MFMA32(av[i], bv[t2], acc[i][t2]);
            }
        }
    }

    // ---- write-out through LDS: per pass 8 output channels x [nci][KH][KW] per co-group
    int nci_valid = a.Cin - ci_base; if (nci_valid > nci) nci_valid = nci;
    const int RL = nci * KHKW;                  // LDS row pitch
    const int RLv = nci_valid * KHKW;           // valid (contiguous in OIHW) run per output channel
    const int ngroups = a.cot / (MS * 32);
    float* out = (a.ksplit > 1 && !a.atomic) ? (a.slabs + (long long)zsplit * a.slab_stride) : a.dw;
    float* Tt = smem;
#pragma unroll
    for (int i = 0; i < MS; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __syncthreads();
            if (lane_ok) {
#pragma unroll
                for (int t2 = 0; t2 < KWT; ++t2) {
                    const int kw = (a.lane_mode == 0) ? (kw0 + t2) : lane_kw;
                    if (kw < a.KW) {
                        const int e = lane_ci * KHKW + kh * a.KW + kw;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) Tt[(cgrp * 8 + rr + 4 * half) * RL + e] = acc[i][t2][4 * q + rr];
                    }
                }
            }
            __syncthreads();
            const int total = ngroups * 8 * RLv;
#pragma unroll 4
            for (int idx = tid; idx < total; idx += nthreads) {
                const int grow = idx / RLv, e = idx - grow * RLv;       // grow = g*8 + row
                const int g = grow >> 3, row = grow & 7;
                const int co = co0 + g * (MS * 32) + i * 32 + q * 8 + row;
                if (co < a.Cout) {
                    const long long o = ((long long)co * a.Cin + ci_base) * KHKW + e;
                    const float v = Tt[grow * RL + e];
                    if (a.atomic) unsafeAtomicAdd(out + o, v);       // tiny dW: coalesced L2 atomics, no slabs
                    else if (a.ksplit > 1) out[o] = v;
                    else out[o] += v;
                }
            }
        }
    }
}

// dw[i] += sum_z slabs[z][i]   (slab z starts at z*stride; stride % 4 == 0; vec: dw is 16-byte aligned)
struct WgradReduceKArgs { float* dw; const float* slabs; long long n; long long stride; int ksplit; int vec; };
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const Twin<WgradReduceKArgs> tw)
{
    const WgradReduceKArgs ka_ = tw.v[blockIdx.z];
    float* __restrict__ dw = ka_.dw;
    const float* __restrict__ slabs = ka_.slabs;
    long long n = ka_.n;
    long long stride = ka_.stride;
    int ksplit = ka_.ksplit;
    int vec = ka_.vec;
    if (!vec) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            float acc = dw[i];
            for (int z = 0; z < ksplit; ++z) acc += slabs[(long long)z * stride + i];
            dw[i] = acc;
        }
        return;
    }
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 acc = reinterpret_cast<float4*>(dw)[i];
        for (int z = 0; z < ksplit; ++z) {
            const float4 v = reinterpret_cast<const float4*>(slabs + (long long)z * stride)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(dw)[i] = acc;
    }
    const long long t = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        float acc = dw[t];
        for (int z = 0; z < ksplit; ++z) acc += slabs[(long long)z * stride + t];
        dw[t] = acc;
    }
}

// ---- small-K weight gradient (1-D trunk at small batch) ---------------------------------------------------
// dW[co][ci][kw] += sum_{row, ow} dY[row][co][ow] * X[row][ci][ow + kw - pw]     (KH == 1, stride 1)
// With only tens of pixels this is an outer-product, bound by the dW read-modify-write (1.5 MB per trunk conv),
// not by FLOPs: one thread per input channel, COB output channels per block, dY broadcast from LDS, dW written
// with coalesced accesses.  Three dependent memory round trips in total (dY stage, X rows, dW RMW).
struct SmallKArgs {
    const float* x; const float* dy; float* dw;
    long long x_sb, x_sc, dy_sb, dy_sc;
    int x_sh, dy_sh;
    int NB, Cin, Cout, OH, OW, pw;
};

// (r3: a thread owns ONE (ci, kw) element of COB consecutive dW rows instead of one ci with its KW taps: the dW read-modify-write -- the
// whole cost of the layer -- is a run of 256 consecutive floats per row and instruction instead of every KW-th float, and x comes through LDS
// with coalesced loads instead of 64 cache lines per load instruction.  5120 x 256 conv1dto2d gradient: 59 -> see DESIGN section 9.)
template <int KW, int COB>
__global__ void __launch_bounds__(256) wgrad_smallk_kernel(const Twin<SmallKArgs> tw)
{
    const SmallKArgs a = tw.v[blockIdx.z];
    extern __shared__ float smk[];                    // dy[COB][npix] | x[channels of the block][npix + 1]
    const int tid = threadIdx.x;
    const int co0 = blockIdx.x * COB;
    const int e0 = blockIdx.y * 256;                  // first (ci, kw) element of this block
    const int rows = a.NB * a.OH;
    const int npix = rows * a.OW, xp = npix + 1;
    float* dys = smk;
    float* xs = smk + COB * npix;
    const int ci0 = e0 / KW;
    int nci = (e0 + 255) / KW - ci0 + 1;
    if (ci0 + nci > a.Cin) nci = a.Cin - ci0;
    for (int i = tid; i < COB * npix; i += 256) {
        const int co = i / npix, p = i - co * npix;
        const int row = p / a.OW, ow = p - row * a.OW;
        const int n = row / a.OH, oh = row - n * a.OH;
        const int cg = co0 + co;
        dys[co * npix + p] = (cg < a.Cout) ? a.dy[(long long)n * a.dy_sb + (long long)cg * a.dy_sc + (long long)oh * a.dy_sh + ow] : 0.f;
    }
    // (eight loads in flight per thread before the first LDS store: a plain load -> store loop is one global round trip per iteration)
    for (int i0 = tid; i0 < nci * npix; i0 += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int c = i / npix, p = i - c * npix;
            const int row = p / a.OW, q = p - row * a.OW;
            const int n = row / a.OH, oh = row - n * a.OH;
            v[u] = (i < nci * npix) ? a.x[(long long)n * a.x_sb + (long long)(ci0 + c) * a.x_sc + (long long)oh * a.x_sh + q] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int c = i / npix, p = i - c * npix;
            if (i < nci * npix) xs[c * xp + p] = v[u];
        }
    }
    __syncthreads();
    const int e = e0 + tid;
    const int ci = e / KW, k = e - ci * KW;
    if (ci >= a.Cin) return;
    const float* xr = xs + (ci - ci0) * xp;
    float acc[COB];
#pragma unroll
    for (int c = 0; c < COB; ++c) acc[c] = 0.f;
    const int sh = k - a.pw;                          // dW[co][ci][k] pairs dy[ow] with x[ow + k - pw]
    for (int row = 0; row < rows; ++row) {
        const float* dr = dys + row * a.OW;
        const float* xq = xr + row * a.OW;
        const int lo = sh < 0 ? -sh : 0, hi = sh > 0 ? a.OW - sh : a.OW;
        for (int ow = lo; ow < hi; ++ow) {
            const float xv = xq[ow + sh];
#pragma unroll
            for (int c = 0; c < COB; ++c) acc[c] += dr[c * npix + ow] * xv;
        }
    }
    // read-modify-write of the COB rows: all loads first (written as `*d += acc` the compiler keeps the rows' round trips in sequence:
    // it cannot prove that the stores do not alias the next row's load)
    float old[COB];
#pragma unroll
    for (int c = 0; c < COB; ++c) old[c] = (co0 + c < a.Cout) ? a.dw[(long long)(co0 + c) * a.Cin * KW + e] : 0.f;
#pragma unroll
    for (int c = 0; c < COB; ++c)
        if (co0 + c < a.Cout) a.dw[(long long)(co0 + c) * a.Cin * KW + e] = old[c] + acc[c];
}

// ---- the same, batched over layers (k = 3, trunk layout [C][B][T4], B*T4 <= 128): block -> (job, 4 output channels, 256 input channels)
struct SmallKBatch { SmallKJob job[MCVC_SMALLK_MAX_JOBS]; int first[MCVC_SMALLK_MAX_JOBS + 1]; int njobs, B, T4; };

constexpr int kSmallKBatchCob = 8;
__global__ void __launch_bounds__(256) wgrad_smallk_batch_kernel(const Twin<SmallKBatch> tw)
{
    const SmallKBatch& bt = tw.v[blockIdx.z];
    constexpr int KW = 3, COB = kSmallKBatchCob;
    extern __shared__ float smk[];
    const int tid = threadIdx.x;
    int j = 0;
    while (j + 1 < bt.njobs && (int)blockIdx.x >= bt.first[j + 1]) ++j;
    const SmallKJob jb = bt.job[j];
    const int local = (int)blockIdx.x - bt.first[j];
    const int e_tiles = (jb.Cin * KW + 255) >> 8;      // blocks along the (ci, kw) elements of a dW row
    const int co0 = (local / e_tiles) * COB;
    const int e0 = (local % e_tiles) * 256;
    const int B = bt.B, T4 = bt.T4;
    const int npix = B * T4, xp = npix + 1;
    float* dys = smk;
    float* xs = smk + COB * npix;
    const int ci0 = e0 / KW;
    int nci = (e0 + 255) / KW - ci0 + 1;
    if (ci0 + nci > jb.Cin) nci = jb.Cin - ci0;
    for (int i = tid; i < COB * npix; i += 256) {
        const int co = i / npix, p = i - co * npix;
        dys[co * npix + p] = (co0 + co < jb.Cout) ? jb.dy[(long long)(co0 + co) * npix + p] : 0.f;
    }
    const int xpitch = (jb.xB > B ? jb.xB : B) * T4;       // (a prefix of a larger forward pass's samples: the channel pitch is that pass's)
    for (int i0 = tid; i0 < nci * npix; i0 += 8 * 256) {   // x rows of the block's channels: consecutive addresses (trunk layout [C][B][T4])
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int c = i / npix, p = i - c * npix;
            v[u] = (i < nci * npix) ? jb.x[(long long)(ci0 + c) * xpitch + p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int c = i / npix, p = i - c * npix;
            if (i < nci * npix) xs[c * xp + p] = v[u];
        }
    }
    __syncthreads();
    const int e = e0 + tid;
    const int ci = e / KW, k = e - ci * KW;
    if (ci >= jb.Cin) return;
    const float* xr = xs + (ci - ci0) * xp;
    float acc[COB];
#pragma unroll
    for (int c = 0; c < COB; ++c) acc[c] = 0.f;
    const int sh = k - 1;
    for (int b = 0; b < B; ++b) {
        const float* dr = dys + b * T4;
        const float* xq = xr + b * T4;
        const int lo = sh < 0 ? 1 : 0, hi = sh > 0 ? T4 - 1 : T4;
        for (int ow = lo; ow < hi; ++ow) {
            const float xv = xq[ow + sh];
#pragma unroll
            for (int c = 0; c < COB; ++c) acc[c] += dr[c * npix + ow] * xv;
        }
    }
    float old[COB];                                   // (all loads of the read-modify-write first: see wgrad_smallk_kernel)
#pragma unroll
    for (int c = 0; c < COB; ++c) old[c] = (co0 + c < jb.Cout) ? jb.dw[(long long)(co0 + c) * jb.Cin * KW + e] : 0.f;
#pragma unroll
    for (int c = 0; c < COB; ++c)
        if (co0 + c < jb.Cout) jb.dw[(long long)(co0 + c) * jb.Cin * KW + e] = old[c] + acc[c];
}

bool mcvc_wgrad_smallk_batch_applies(int B, int T4) { return B >= 1 && T4 >= 1 && (long long)B * T4 <= 128; }

int mcvc_wgrad_smallk_batch_launch(const SmallKJob* jobs, int njobs, int B, int T4, hipStream_t s)
{
    if (njobs < 1 || njobs > MCVC_SMALLK_MAX_JOBS || !mcvc_wgrad_smallk_batch_applies(B, T4)) return MCVC_ERR_INVALID;
    SmallKBatch bt{};
    int total = 0;
    double flops = 0.0, bytes = 0.0;
    const double px = (double)B * T4;
    for (int j = 0; j < njobs; ++j) {
        bt.job[j] = jobs[j];
        bt.first[j] = total;
        total += cdiv_i(jobs[j].Cout, kSmallKBatchCob) * cdiv_i(jobs[j].Cin * 3, 256);
        flops += 2.0 * px * jobs[j].Cout * jobs[j].Cin * 3;
        bytes += 4.0 * (2.0 * jobs[j].Cout * jobs[j].Cin * 3 + px * (jobs[j].Cin + jobs[j].Cout));
    }
    bt.first[njobs] = total; bt.njobs = njobs; bt.B = B; bt.T4 = T4;
    TraceScope ts(K_WGRAD_SMALLK, s, flops, bytes);
    const size_t lds = (size_t)(kSmallKBatchCob * B * T4 + (256 / 3 + 2) * (B * T4 + 1)) * sizeof(float);
    mcvc_launch(wgrad_smallk_batch_kernel, dim3((unsigned)total), dim3(256), lds, s, bt);
    return (int)hipGetLastError();
}

namespace {
template <int MS, int KWT, int MAXT, int MINW = 1>
static hipError_t launch_wgrad(const WgradArgs& a, dim3 grid, int nwaves, size_t lds, hipStream_t s)
{
    if (64 * nwaves > MAXT) return hipErrorInvalidValue;
    auto kern = conv_wgrad_kernel<MS, KWT, MAXT, MINW>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int kind = (MS == 2 && KWT == 5) ? K_WGRAD_2x5 : (MS == 1 && KWT == 5) ? K_WGRAD_1x5 : (MS == 2 && KWT == 3) ? K_WGRAD_2x3
                   : (MS == 1 && KWT == 3) ? K_WGRAD_1x3 : (MS == 4) ? K_WGRAD_4x1 : K_WGRAD_1x1;
    const double px = (double)a.NB * a.OH * a.OW;
    TraceScope ts(kind, s, 2.0 * px * a.Cout * a.Cin * a.KH * a.KW,
                  4.0 * ((double)a.NB * a.Cin * a.H * a.W + px * a.Cout + (double)a.Cout * a.Cin * a.KH * a.KW * (a.ksplit > 1 ? a.ksplit : 2)));
    mcvc_launch(kern, grid, dim3(64 * nwaves), lds, s, a);
    return hipGetLastError();
}

// planner knobs (tools/wgrad_tune.py): read once, or on every call when MCVC_WGRAD_TUNE=1 was set at load time
static int wgrad_knob(const char* name, int dflt)
{
    static const int tune = env_int("MCVC_WGRAD_TUNE", 0);
    if (tune) return env_int(name, dflt);
    return dflt;
}
static int wgrad_knob_ms() { return wgrad_knob("MCVC_WGRAD_MS", 0); }
static int wgrad_knob_waves() { return wgrad_knob("MCVC_WGRAD_WAVES", 0); }
static int wgrad_knob_lds_floats() { return wgrad_knob("MCVC_WGRAD_LDS_KB", 78) * 256; }
static int wgrad_knob_ksplit() { return wgrad_knob("MCVC_WGRAD_KSPLIT", 0); }

struct WgradPlan { WgradArgs a; dim3 grid; int MS, KWT, nwaves; size_t lds; };

// force_slabs: never use the atomic tiny-dW path (deterministic mode)
static bool plan_wgrad(const ConvProblem& p, int NB, long long slab_cap_floats, WgradPlan* out, bool force_slabs)
{
    WgradPlan pl{};
    WgradArgs& a = pl.a;
    a.NB = NB; a.Cin = p.Cin; a.H = p.H; a.W = p.W; a.Cout = p.Cout; a.OH = p.OH; a.OW = p.OW;
    a.KH = p.KH; a.KW = p.KW; a.stride = p.stride; a.pad_h = p.pad_h; a.pad_w = p.pad_w;
    a.dw_floats = (long long)p.Cout * p.Cin * p.KH * p.KW;
    a.slab_stride = (a.dw_floats + 3) & ~3LL;
    // Tile shape.  MS = 1 (80 / 48 accumulator registers) everywhere: measured on gfx950 the smaller register footprint
    // (more resident waves to hide the LDS reads) beats the better MFMA : ds_read ratio of MS = 2.
    int MS, KWT, wave_cap = 8;
    a.lane_mode = (p.Cin * p.KW <= 32 && p.Cin <= 4) ? 1 : 0;
    if (a.lane_mode == 1) { MS = (p.Cout >= 128) ? 4 : 1; KWT = 1; a.nkwg = 1; }
    else if (p.KW % 5 == 0) { KWT = 5; a.nkwg = p.KW / 5; MS = 1; wave_cap = 10; }      // 2 co-groups x 5 kernel rows
    else if (p.KW == 3) { KWT = 3; a.nkwg = 1; MS = 1; }
    else if (p.KW == 1) { KWT = 1; a.nkwg = 1; MS = (p.Cout >= 128) ? 4 : 1; }
    else return false;
    const int tasks = p.KH * a.nkwg;
    if (tasks > 16) return false;
    if (tasks == 1) wave_cap = 4;
    int groups = 1;
    if (a.lane_mode == 0 && wgrad_knob_ms() > 0 && p.Cout >= 32 * wgrad_knob_ms() && KWT > 1) MS = wgrad_knob_ms();
    if (wgrad_knob_waves() > 0) wave_cap = wgrad_knob_waves();
    const int max_groups = cdiv_i(p.Cout, 32 * MS);
    while (groups * 2 * tasks <= wave_cap && groups * 2 <= max_groups && groups * 2 * MS * 32 <= 256) groups *= 2;
    pl.nwaves = groups * tasks;
    pl.MS = MS; pl.KWT = KWT;
    a.cot = groups * MS * 32;

    int tow = 32;
    while (tow > 2 && tow / 2 >= p.OW) tow /= 2;      // smallest power of two >= OW, capped at 32
    a.tow = tow;
    a.tiles_w = cdiv_i(p.OW, tow);
    const int nci = (a.lane_mode == 0) ? 32 : p.Cin;
    const int khkw = p.KH * p.KW;
    const int tfloats = groups * 8 * nci * khkw;      // write-out transpose tile
    const int ci_tiles = (a.lane_mode == 0) ? cdiv_i(p.Cin, 32) : 1;
    const int co_tiles = cdiv_i(p.Cout, a.cot);
    const int base = ci_tiles * co_tiles;

    // K split: the sweep in tools/wgrad_tune.py puts the optimum of every layer of the network at "one workgroup per CU"
    // (256): fewer leaves CUs idle, more multiplies the dW-sized slab traffic.  Tiles are made as tall as LDS allows but
    // no taller than what still yields `want` pixel tiles, and balanced (toh = ceil(OH / tiles_h)).
    int want = (base >= 256) ? 1 : cdiv_i(256, base);
    // tiny dW (edge layers: a few thousand outputs over thousands of pixels): K-split workgroups add straight into dW with
    // coalesced atomics -- the slab round trip and the reduce launch would cost more than the layer
    const bool tiny = !force_slabs && a.dw_floats <= (long long)wgrad_knob("MCVC_WGRAD_ATOMIC_BELOW", 65536);
    if (tiny) { if (want > 128) want = 128; }
    else {
        long long cap = a.slab_stride > 0 ? slab_cap_floats / a.slab_stride : 0;
        const long long hard = 32;
        if (cap > hard) cap = hard;
        if (want > cap) want = (int)cap;
        if (want < 2) want = 1;
    }
    if (wgrad_knob_ksplit() > 0) {
        want = wgrad_knob_ksplit();
        const long long cap = a.slab_stride > 0 ? slab_cap_floats / a.slab_stride : 0;
        if (!tiny && want > cap) want = cap < 2 ? 1 : (int)cap;
    }
    a.atomic = tiny ? 1 : 0;
    // two workgroups per CU can overlap each other's staging / write-out only if there are that many of them
    const int lds_budget = wgrad_knob("MCVC_WGRAD_LDS_KB", ((long long)base * want >= 512) ? 78 : 156) * 256;
    int toh_max = 1;
    for (int cand = 1; cand <= p.OH && cand <= 32; ++cand) {
        const int PH = (cand - 1) * p.stride + p.KH, PW = (tow - 1) * p.stride + p.KW;
        const int plane = (PH * PW) | 1;
        const int pitch_a = (cand * tow) | 1;
        if (a.cot * pitch_a + nci * plane + 64 <= lds_budget && PH * PW < 65536) toh_max = cand; else break;
    }
    int tiles_h = cdiv_i(p.OH, toh_max);
    {
        const int need = cdiv_i(want, NB * a.tiles_w);
        if (need > tiles_h) tiles_h = need;
        if (tiles_h > p.OH) tiles_h = p.OH;
        // prefer a tile count that the K split divides evenly (a few extra, shorter tiles are cheaper than one idle round)
        int best = tiles_h, best_cost = 1 << 30;
        for (int th = tiles_h; th <= p.OH && th < tiles_h + 4; ++th) {
            const int toh_c = cdiv_i(p.OH, th), it = NB * cdiv_i(p.OH, toh_c) * a.tiles_w;
            const int ks = it < want ? it : want;
            const int cost = cdiv_i(it, ks) * (toh_c + 1);          // rows walked by the busiest workgroup (+1: per-tile overhead)
            if (cost < best_cost) { best_cost = cost; best = th; }
        }
        tiles_h = best;
    }
    const int toh = cdiv_i(p.OH, tiles_h);
    a.toh = toh;
    a.PH = (toh - 1) * p.stride + p.KH;
    a.PW = (tow - 1) * p.stride + p.KW;
    a.PWp = a.PW;
    a.plane = (a.PH * a.PWp) | 1;
    a.pitch_a = (toh * tow) | 1;
    a.tiles_h = cdiv_i(p.OH, toh);
    int lds_floats = a.cot * a.pitch_a + nci * a.plane + 64;
    if (tfloats > lds_floats) lds_floats = tfloats;
    pl.lds = (size_t)lds_floats * sizeof(float);
    if (pl.lds > 160 * 1024) return false;

    const int items = NB * a.tiles_h * a.tiles_w;
    int ksplit = want < items ? want : items;
    if (ksplit < 1) ksplit = 1;
    a.ksplit = ksplit; a.ci_tiles = ci_tiles;
    pl.grid = dim3((unsigned)(ci_tiles * co_tiles * ksplit));
    if (wgrad_knob("MCVC_WGRAD_VERBOSE", 0))
        fprintf(stderr, "[wgrad plan] Cin=%d Cout=%d k=%dx%d s=%d NB=%d out=%dx%d | MS=%d KWT=%d waves=%d cot=%d tow=%d toh=%d items=%d ksplit=%d grid=%d lds=%zu\n",
                p.Cin, p.Cout, p.KH, p.KW, p.stride, NB, p.OH, p.OW, MS, KWT, pl.nwaves, a.cot, a.tow, a.toh, items, ksplit, (int)pl.grid.x, pl.lds);
    *out = pl;
    return true;
}
}  // namespace

static bool smallk_applies(const ConvProblem& p, int NB);
long long mcvc_wgrad_plan_slab_floats(const ConvProblem& p, int NB)
{
    if (smallk_applies(p, NB)) return 0;
    // sized for the deterministic (slab-only) plan as well, so the mode can be switched without re-sizing workspaces
    long long need = 0;
    for (int det = 0; det < 2; ++det) {
        WgradPlan pl;
        if (!plan_wgrad(p, NB, 1LL << 40, &pl, det != 0)) return -1;
        const long long n = (pl.a.ksplit > 1 && !pl.a.atomic) ? (long long)pl.a.ksplit * pl.a.slab_stride : 0;
        if (n > need) need = n;
    }
    return need;
}

static bool smallk_applies(const ConvProblem& p, int NB)
{
    return p.KH == 1 && p.stride == 1 && (p.KW == 1 || p.KW == 3) && p.pad_h == 0 && (long long)NB * p.OH * p.OW <= 128 && p.OW == p.W;
}

int mcvc_wgrad_launch(const ConvProblem& p, int NB, const WgradIO& io, float* dw, float* slabs, long long slab_cap_floats, hipStream_t s)
{
    if (smallk_applies(p, NB)) {
        SmallKArgs k{};
        k.x = io.x; k.dy = io.dy; k.dw = dw;
        k.x_sb = io.x_sb; k.x_sc = io.x_sc; k.x_sh = io.x_sh;
        k.dy_sb = io.dy_sb; k.dy_sc = io.dy_sc; k.dy_sh = io.dy_sh;
        k.NB = NB; k.Cin = p.Cin; k.Cout = p.Cout; k.OH = p.OH; k.OW = p.OW; k.pw = p.pad_w;
        // output channels per workgroup: every workgroup stages the x rows of its 256 (ci, kw) elements, so layers with enough workgroups
        // (conv1dto2d 5120 x 256, conv2dto1d 256 x 5120) take 16 rows per workgroup instead of 4 (a quarter of the x re-reads)
        const int COB = (p.KW == 1 && cdiv_i(p.Cout, 16) * cdiv_i(p.Cin, 256) >= 256) ? 16 : 4;
        dim3 grid((unsigned)cdiv_i(p.Cout, COB), (unsigned)cdiv_i(p.Cin * p.KW, 256));
        const double px = (double)NB * p.OH * p.OW;
        TraceScope ts(K_WGRAD_SMALLK, s, 2.0 * px * p.Cout * p.Cin * p.KW, 4.0 * (2.0 * p.Cout * p.Cin * p.KW + px * (p.Cin + p.Cout)));
        const int npix = NB * p.OH * p.OW;
        const size_t lds = (size_t)(COB * npix + (256 / p.KW + 2) * (npix + 1)) * sizeof(float);       // <= 64 KB at 128 pixels
        if (p.KW == 3) mcvc_launch((wgrad_smallk_kernel<3, 4>), grid, dim3(256), lds, s, k);
        else if (COB == 16) mcvc_launch((wgrad_smallk_kernel<1, 16>), grid, dim3(256), lds, s, k);
        else mcvc_launch((wgrad_smallk_kernel<1, 4>), grid, dim3(256), lds, s, k);
        return (int)hipGetLastError();
    }
    WgradPlan pl;
    if (!plan_wgrad(p, NB, slabs ? slab_cap_floats : 0, &pl, mcvc_deterministic() != 0)) return MCVC_ERR_INVALID;
    WgradArgs& a = pl.a;
    a.x = io.x; a.x_sb = io.x_sb; a.x_sc = io.x_sc; a.x_sh = io.x_sh;
    a.dy = io.dy; a.dy_sb = io.dy_sb; a.dy_sc = io.dy_sc; a.dy_sh = io.dy_sh;
    a.dw = dw; a.slabs = slabs;
    hipError_t e;
    const int MS = pl.MS, KWT = pl.KWT;
    // MAXT bounds the register allocator: 512 threads -> up to 256 VGPRs (accumulator-heavy shapes)
    if (MS == 2 && KWT == 5) e = launch_wgrad<2, 5, 512>(a, pl.grid, pl.nwaves, pl.lds, s);
    else if (MS == 1 && KWT == 5) e = launch_wgrad<1, 5, 1024>(a, pl.grid, pl.nwaves, pl.lds, s);
    else if (MS == 2 && KWT == 3) e = launch_wgrad<2, 3, 512>(a, pl.grid, pl.nwaves, pl.lds, s);
    else if (MS == 1 && KWT == 3) e = launch_wgrad<1, 3, 1024>(a, pl.grid, pl.nwaves, pl.lds, s);
    else if (MS == 4 && KWT == 1) e = launch_wgrad<4, 1, 512>(a, pl.grid, pl.nwaves, pl.lds, s);
    else e = launch_wgrad<1, 1, 1024>(a, pl.grid, pl.nwaves, pl.lds, s);
    if (e != hipSuccess) return (int)e;
    if (a.ksplit > 1 && !a.atomic) {
        long long b = cdiv_ll(a.dw_floats >> 2, 256);
        if (b > 2048) b = 2048;
        if (b < 1) b = 1;
        TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * (double)a.dw_floats * (a.ksplit + 2));
        const int vec = (((uintptr_t)dw | (uintptr_t)slabs) & 15) == 0;
        mcvc_launch(wgrad_reduce_kernel, dim3((unsigned)b), dim3(256), 0, s, WgradReduceKArgs{dw, slabs, a.dw_floats, a.slab_stride, a.ksplit, vec});
        e = hipGetLastError();
    }
    return (int)e;
}
