// Fused 1-D trunk layer for small batch (N = B * T/4 <= 32 output columns):
//   conv1d (k = 1 or 3) + bias + InstanceNorm1d(affine) + {gated GLU | residual add | nothing}   -- ONE launch.
//
// Replaces, for the generator's residual trunk (reference mask_cyclegan_vc/model.py:47-76, 142-189, 254-267), the
// generic direct-conv kernel + split-K slabs + separate norm kernel.  At bs=1 a trunk conv is 0.013-0.05 GFLOP over
// 1.5-3 MB of weights: pure weight streaming and latency.  The timeline analysis (tools/rocpd_timeline.py) showed
// these 52-workgroup launches alone on the chip for ~20 % of the step, so the design goal here is the fewest possible
// dependent memory round trips, not FLOP/s:
//   * GEMM view  out[m][n] = sum_k A[m][k] * X[k][n],  m = output channel, n = (b, t), k = (ci, kw).
//     One workgroup owns 16 output rows (GLU: 8 value + the 8 matching gate rows) and ALL N <= 32 columns = one or two
//     16x16 fp32 MFMA accumulators per wave; its 8 waves split K and are summed through LDS -> no inter-block split
//     (16-row tiles rather than 32: twice the workgroups, i.e. twice the DRAM requests in flight, and 64-byte instead
//     of 32-byte contiguous pieces per weight row and load instruction).
//   * A (weights) is read straight from the OIHW parameter tensor ([m][k], k contiguous): 16-byte loads per lane, ALL
//     of a wave's loads issued before anything else, each float4 feeding four v_mfma_f32_16x16x4_f32 steps.
//     No packed copy, no LDS staging.
//   * X (at most 512 x 32 activations, with zero halo) is staged once in LDS; the MFMA B operand is a shifted read.
//   * Epilogue in LDS: bias, pre-norm store (backward needs it), per-(row, b) two-pass statistics, affine, GLU /
//     residual, strided store (which also performs the reference's view(B,256,20,-1) when writing NCHW).
//   * K = 5120 (conv2dto1d and the data-gradient of conv1dto2d) does not fit one workgroup's LDS: K-split workgroups write
//     private slabs (split 0 adds the bias) that the following norm launch sums -- no atomics, no zero-fill launch.
// The same kernel with mode 0 is the data-gradient of these layers (weights transposed+flipped by pack_trunk_t).
//
// r4 -- LDS layout and k mapping (SQ_LDS_BANK_CONFLICT was 31-46 % of the LDS cycles of these kernels):
//   * a staged row is [4 zeros | T4 values] (k = 3; k = 1: the T4 values alone): the values are 16-byte aligned, so staging is one
//     ds_write_b128 per float4 instead of four 4-way-conflicting ds_write_b32; the zeros in front of row r+1 are the right halo of row r;
//   * the four k of one 16x16x4 MFMA are no longer 4 apart in the weight row (mixed (ci, kw) pairs whose LDS addresses differ by
//     RS+1 or 2*RS-2 -- no pitch separates both) but lane-quarter kq owns WHOLE channels: kq's float4s of a super-group are consecutive
//     (k = 4*(NQ*kq + q) + j), so in every MFMA step the quarters kq and kq+1 read the same tap of channels 4 (k = 3) or 8 (k = 1)
//     apart, i.e. 4*RS or 8*RS floats apart; the channel pitch RS is padded to 4 mod 8 (k = 3) / 2 mod 4 (k = 1), which puts the two
//     16-lane windows of a ds_read_b32 lane group exactly 16 banks apart: conflict-free at T4 = 16;
//   * up to 64 columns (one to four 16-column accumulators per wave): three samples of 64 frames per pass (the trainer's merged forward).
#include "mcvc_common.h"
#include "trace.h"
#include "launch.h"
#include "trunk.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int kTrunkWaves = 8;                 // K is split over the waves of ONE workgroup (2 per SIMD)
constexpr int kTrunkThreads = 64 * kTrunkWaves;

// staged-row geometry (see the header): data offset inside a row, row pitch, channel pitch
__host__ __device__ inline int trunk_doff(int KW) { return KW == 3 ? 4 : 0; }
__host__ __device__ inline int trunk_tp(int T4, int KW) { return T4 + trunk_doff(KW); }
__host__ __device__ inline int trunk_rs(int B, int T4, int KW)
{
    int rs = B * trunk_tp(T4, KW);
    if (KW == 3) { while ((rs & 7) != 4) ++rs; }      // quarters 4 channels apart -> 16 banks apart
    else { while ((rs & 3) != 2) ++rs; }              // quarters 8 channels apart -> 16 banks apart
    return rs;
}
__host__ __device__ inline int trunk_na(int N) { return (N + 15) >> 4; }
// epilogue scratch of one workgroup: red[waves][NA][4][64] | tile[16][16 NA + 1] | sstat[16][8][2] | 16 spare (flag)
// (red rows are kRedPitch = 80 floats apart, not 64: the reducing threads of a 32-lane group read two accumulator registers r, r + 1 of 16
//  lanes each -- 64 apart they would share their 16 banks, 80 apart they sit 16 banks apart)
constexpr int kRedPitch = 80;
__host__ __device__ inline int trunk_epi_floats(int NA) { return kTrunkWaves * NA * 4 * kRedPitch + 16 * (16 * NA + 1) + 16 * 8 * 2 + 16; }

// zero every staged slot that holds no value: the 4 leading zeros of each row (k = 3), the pad behind the last row of a channel, and
// 4 floats behind the last channel (right halo of the very last row / the idle lanes' zero slot)
__device__ __forceinline__ void trunk_zero_gaps(float* Xs, int nch, int B, int TP, int RS, int DOFF, int tid, int nthreads)
{
    const int per = B * DOFF + (RS - B * TP);
    for (int i = tid; i < nch * per + 4; i += nthreads) {
        const int ci = per > 0 ? i / per : nch, j = i - ci * per;
        int pos;
        if (ci >= nch) pos = j;                                        // (the 4 floats behind the last channel)
        else if (j < B * DOFF) { const int b = j / (DOFF > 0 ? DOFF : 1), o = j - b * DOFF; pos = b * TP + o; }
        else pos = B * TP + (j - B * DOFF);
        Xs[ci * RS + pos] = 0.f;
    }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// 16x16x4 fp32 MFMA: lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; D register r is row 4*(l >> 4) + r,
// column l & 15.  A float4 of consecutive k per lane feeds four MFMAs (any 4 distinct k per instruction are fine as long as
// the B operand uses the same ones).
template <int KW, int NA, bool PRE>
__global__ void __launch_bounds__(kTrunkThreads) trunk_layer_kernel(const Twin<TrunkArgs> tw)
{
    const TrunkArgs a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int DOFF = (KW == 3) ? 4 : 0;
    const int TP = a.T4 + DOFF;                   // staged row: [4 zeros | T4 values] (k = 3)
    const int RS = trunk_rs(a.B, a.T4, KW);       // channel pitch (padded: see the header)
    constexpr int TLP = 16 * NA + 1;              // pitch of the epilogue tile
    const int glu = (a.mode == TRUNK_IN_GLU);
    const int rows_per_block = glu ? 8 : 16;
    const int r0 = blockIdx.x * rows_per_block;
    // K range of this workgroup (gridDim.y K-splits, accumulate mode only) and of this wave
    const int kblk = a.K / gridDim.y;
    const int k_count = kblk / kTrunkWaves;       // multiple of the super-group (checked on the host)
    const int k_begin = blockIdx.y * kblk + wave * k_count;
    const int ci_begin = (blockIdx.y * kblk) / KW;
    const int ci_count = kblk / KW;

    // A row of this lane
    const float* arow;
    {
        const int i = l15;
        if (glu) arow = (i < 8) ? (a.a0 + (long long)(r0 + i) * a.K) : (a.a1 + (long long)(r0 + i - 8) * a.K);
        else arow = a.a0 + (long long)(r0 + i) * a.K;
    }
    // One super-group = NQ float4 of weights per lane = GK = 16*NQ consecutive k per wave = whole input channels, so the
    // (ci, kw) pattern repeats and the LDS offsets of its 4*NQ MFMA steps are loop-invariant registers.  Lane quarter kq owns
    // the float4s NQ*kq .. NQ*kq + NQ-1 of the super-group (whole channels: see the header).  Up to CH super-groups (all of
    // them for the generator's shapes) are requested before anything else happens: at this size the layer is one DRAM latency,
    // not a bandwidth problem.
    constexpr int NQ = (KW == 3) ? 3 : 2;
    constexpr int GK = 16 * NQ;
    constexpr int CH = 4;
    const int sgroups = k_count / GK;
    const float* ap = arow + k_begin + 4 * NQ * kq;
    float4 wb[CH][NQ];
#define TRUNK_LOAD_CHUNK(sg0)                                                                                       \
    {                                                                                                              \
        _Pragma("unroll") for (int c = 0; c < CH; ++c) {                                                           \
            const int sg = (sg0) + c;                                                                              \
            const float* pp = ap + (long long)(sg < sgroups ? sg : 0) * GK;                                        \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) wb[c][q] = *reinterpret_cast<const float4*>(pp + 4 * q); \
        }                                                                                                          \
    }
    TRUNK_LOAD_CHUNK(0);

    // ---- stage X[ci][b][t] (this block's channel slice)
    if constexpr (PRE) {
        // fused InstanceNorm backward (see trunk.h).  Work item = a quarter of one (channel, sample) row: four adjacent lanes own a
        // row, each holds ceil(T4/4) <= 8 consecutive elements, the two row sums are width-4 shuffles.  d(gamma), d(beta) of a channel
        // are summed over its samples in a fixed order through a small LDS table (deterministic, no atomics).
        const int C = a.pre_C;
        const bool pglu = (a.pre == 2);
        const int Cx = pglu ? 2 * C : C;
        const int PB = a.pre_xB > 0 ? a.pre_xB : a.B;         // samples per channel of the forward pass's tensor (prefix backward: > B)
        const bool out = (blockIdx.x == 0);
        const float invT = 1.0f / (float)a.T4;
        const int E = (a.T4 + 3) >> 2;                        // elements per lane (<= 8)
        float* rsum = smem + ci_count * RS + 4;               // [ci_count][B][2] row sums (s1, s2) -- behind the staged slice
        const int items = ci_count * a.B * 4;
        for (int it0 = 0; it0 < items; it0 += kTrunkThreads) {
            const int item = it0 + tid;
            const bool live = item < items;                   // (whole 4-lane groups are live or dead together: items % 4 == 0)
            const int q4 = item & 3, row = live ? (item >> 2) : 0;
            const int ci = row / a.B, b = row - ci * a.B;
            const int cx = ci_begin + ci;
            const bool gate = pglu && cx >= C;
            const int c = gate ? cx - C : cx;
            const float g0 = a.pre_gamma0[c], b0 = a.pre_beta0[c];
            const float g1 = pglu ? a.pre_gamma1[c] : 0.f, b1 = pglu ? a.pre_beta1[c] : 0.f;
            const float* st = a.pre_stats + (long long)b * Cx * 2;
            const float m0 = st[2 * c], r0 = st[2 * c + 1];
            const float m1 = pglu ? st[2 * (c + C)] : 0.f, r1 = pglu ? st[2 * (c + C) + 1] : 1.f;
            const int t0 = q4 * E;
            const float* dyr = a.x + (long long)c * a.x_sc + (long long)b * a.x_sb + t0;
            const float* x0r = a.pre_x + ((long long)c * PB + b) * a.T4 + t0;
            const float* x1r = a.pre_x + ((long long)(c + C) * PB + b) * a.T4 + t0;
            float vd[8], v0[8], v1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = live && e < E && (t0 + e) < a.T4;
                vd[e] = ok ? dyr[e] : 0.f; v0[e] = ok ? x0r[e] : 0.f; v1[e] = (ok && pglu) ? x1r[e] : 0.f;
            }
            float dzv[8], xhv[8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = e < E && (t0 + e) < a.T4;
                const float xh0 = (v0[e] - m0) * r0;
                float dz, xh;
                if (pglu) {
                    const float xh1 = (v1[e] - m1) * r1;
                    const float sg = sigmoidf_(xh1 * g1 + b1);
                    if (gate) { dz = vd[e] * (xh0 * g0 + b0) * sg * (1.0f - sg); xh = xh1; }
                    else { dz = vd[e] * sg; xh = xh0; }
                } else { dz = vd[e]; xh = xh0; }
                dzv[e] = ok ? dz : 0.f; xhv[e] = ok ? xh : 0.f;
                s1 += dzv[e]; s2 += dzv[e] * xhv[e];
            }
            s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
            s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
            if (live) {
                const float gr = gate ? g1 * r1 : g0 * r0;
                float* xrow = smem + ci * RS + b * TP + DOFF;
                float* od = out ? (a.pre_out + ((long long)cx * a.B + b) * a.T4 + t0) : nullptr;
                if (a.T4 == 16 && KW == 3) {          // (the trainer's shape: four elements per lane = one 16-byte store, LDS and global)
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gr * (dzv[e] - s1 * invT - xhv[e] * (s2 * invT));
                    *reinterpret_cast<float4*>(xrow + t0) = make_float4(o[0], o[1], o[2], o[3]);
                    if (od) *reinterpret_cast<float4*>(od) = make_float4(o[0], o[1], o[2], o[3]);
                } else
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e < E && (t0 + e) < a.T4) {
                        const float dxv = gr * (dzv[e] - s1 * invT - xhv[e] * (s2 * invT));
                        xrow[t0 + e] = dxv;
                        if (od) od[e] = dxv;
                    }
                }
                if (q4 == 0) { rsum[(ci * a.B + b) * 2] = s1; rsum[(ci * a.B + b) * 2 + 1] = s2; }
            }
        }
        trunk_zero_gaps(smem, ci_count, a.B, TP, RS, DOFF, tid, kTrunkThreads);
        if (out) {
            __syncthreads();
            for (int ci = tid; ci < ci_count; ci += kTrunkThreads) {
                const int cx = ci_begin + ci;
                const bool gate = pglu && cx >= C;
                const int c = gate ? cx - C : cx;
                float dgam = 0.f, dbet = 0.f;
                for (int b = 0; b < a.B; ++b) { dbet += rsum[(ci * a.B + b) * 2]; dgam += rsum[(ci * a.B + b) * 2 + 1]; }
                float* dg = gate ? a.pre_dgamma1 : a.pre_dgamma0;
                float* db = gate ? a.pre_dbeta1 : a.pre_dbeta0;
                if (dg) dg[c] += dgam;
                if (db) db[c] += dbet;
            }
        }
    } else {
        const float* xsrc = a.x + (long long)ci_begin * a.x_sc;
        const bool fast = (a.x_sb == a.T4) && (a.x_sc == (long long)a.B * a.T4) && ((a.T4 & 3) == 0) &&
                          ((reinterpret_cast<unsigned long long>(xsrc) & 15ull) == 0);
        if (fast) {
            const int nf4 = (ci_count * a.B * a.T4) >> 2;
            for (int f0 = 0; f0 < nf4; f0 += 4 * kTrunkThreads) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * kTrunkThreads + tid;
                    v[u] = (f < nf4) ? *reinterpret_cast<const float4*>(xsrc + 4ll * f) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * kTrunkThreads + tid;
                    if (f < nf4) {
                        const int e = 4 * f;
                        const int r = e / a.T4, t = e - r * a.T4;
                        const int ci = r / a.B, b = r - ci * a.B;
                        float* d = smem + ci * RS + b * TP + DOFF + t;
                        if constexpr (KW == 3) *reinterpret_cast<float4*>(d) = v[u];          // (RS, TP, DOFF, t multiples of 4: 16-byte aligned)
                        else { *reinterpret_cast<float2*>(d) = make_float2(v[u].x, v[u].y); *reinterpret_cast<float2*>(d + 2) = make_float2(v[u].z, v[u].w); }
                    }
                }
            }
        } else {
            for (int i = tid; i < ci_count * a.B * a.T4; i += kTrunkThreads) {
                const int r = i / a.T4, t = i - r * a.T4;
                const int ci = r / a.B, b = r - ci * a.B;
                smem[ci * RS + b * TP + DOFF + t] = xsrc[(long long)ci * a.x_sc + (long long)b * a.x_sb + t];
            }
        }
        trunk_zero_gaps(smem, ci_count, a.B, TP, RS, DOFF, tid, kTrunkThreads);
    }
    // B columns of this lane: n = l15 (+16 per further accumulator) -> (b, t); idle lanes read a zero slot
    int off[NA][NQ][4];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        const int n = l15 + 16 * h;
        int xcol = a.B * TP, live = 0;
        if (n < a.N) { const int b = n / a.T4, t = n - b * a.T4; xcol = b * TP + DOFF - (KW - 1) / 2 + t; live = 1; }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = 4 * (NQ * kq + q) + j;
                off[h][q][j] = (kr / KW) * RS + xcol + (live ? (kr % KW) : 0);
            }
    }
    f32x4 acc[NA];
#pragma unroll
    for (int h = 0; h < NA; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // ---- main loop
    {
        const float* xs = smem + (k_begin / KW - ci_begin) * RS;
        for (int sg0 = 0; sg0 < sgroups; sg0 += CH) {
            if (sg0 > 0) TRUNK_LOAD_CHUNK(sg0);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (sg0 + c < sgroups) {
                    // all operand reads of the super-group first, pinned in front of its MFMAs: left alone the scheduler emits
                    // ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per product, i.e. one LDS latency (~150 cycles) per 32-cycle MFMA
                    float xv[NQ][4][NA];
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int h = 0; h < NA; ++h) xv[q][j][h] = xs[off[h][q][j]];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float av[4] = {wb[c][q].x, wb[c][q].y, wb[c][q].z, wb[c][q].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int h = 0; h < NA; ++h) acc[h] = MFMA16(av[j], xv[q][j][h], acc[h]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, NQ * 4 * NA, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NQ * 4 * NA, 0);
                    xs += (GK / KW) * RS;
                }
            }
        }
#undef TRUNK_LOAD_CHUNK
    }
    __syncthreads();                               // everyone is done reading X from LDS

    // ---- cross-wave K reduction through LDS: red[wave][h][reg][lane]
    float* red = smem;
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NA + h) * 4 + r) * kRedPitch + lane] = acc[h][r];
    __syncthreads();
    float* tile = smem + kTrunkWaves * NA * 4 * kRedPitch;  // [16 rows][TLP]
    float* sstat = tile + 16 * TLP;                // [16 rows][B][2]
    for (int e = tid; e < NA * 256; e += kTrunkThreads) {
        const int h = e >> 8, r = (e >> 6) & 3, ln = e & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kTrunkWaves; ++w) v += red[((w * NA + h) * 4 + r) * kRedPitch + ln];
        const int row = 4 * (ln >> 4) + r, col = (ln & 15) + 16 * h;
        tile[row * TLP + col] = v;
    }
    __syncthreads();

    // row -> output channel(s)
    auto row_cx = [&](int row) { return glu ? ((row < 8) ? (r0 + row) : (a.M + r0 + row - 8)) : (r0 + row); };

    if (a.mode == TRUNK_PLAIN) {                    // data-gradient / K-split forward: store, accumulate, or private slab
        const bool slab = (gridDim.y > 1) && !a.accumulate;
        float* base = (slab && blockIdx.y > 0) ? (a.slabs + (long long)(blockIdx.y - 1) * a.slab_stride) : a.conv_out;
        if (a.slab_all) base = a.slabs + (long long)blockIdx.y * a.slab_stride;
        for (int e = tid; e < 16 * a.N; e += kTrunkThreads) {
            const int row = e / a.N, nn = e - row * a.N;
            const int b = nn / a.T4, t = nn - b * a.T4;
            float* dst = base + (long long)row_cx(row) * a.c_sc + (long long)b * a.c_sb + t;
            float v = tile[row * TLP + nn];
            if (a.bias0 && blockIdx.y == 0) v += a.bias0[r0 + row];
            if (a.slab_all) *dst = v;
            else if (a.accumulate) { if (gridDim.y > 1) unsafeAtomicAdd(dst, v); else *dst += v; }
            else *dst = v;
        }
        return;
    }
    // ---- bias + pre-norm store
    for (int e = tid; e < 16 * a.N; e += kTrunkThreads) {
        const int row = e / a.N, nn = e - row * a.N;
        const int cx = row_cx(row);
        const float bias = glu ? ((row < 8) ? a.bias0[r0 + row] : a.bias1[r0 + row - 8]) : a.bias0[r0 + row];
        const float v = tile[row * TLP + nn] + bias;
        tile[row * TLP + nn] = v;
        const int b = nn / a.T4, t = nn - b * a.T4;
        a.conv_out[(long long)cx * a.c_sc + (long long)b * a.c_sb + t] = v;
    }
    __syncthreads();
    // ---- statistics per (row, b): two-pass over T4 (<= 32) elements, one thread each
    if (tid < 16 * a.B) {
        const int row = tid / a.B, b = tid - row * a.B;
        const float* p = tile + row * TLP + b * a.T4;
        float s = 0.f;
        for (int t = 0; t < a.T4; ++t) s += p[t];
        const float mean = s / (float)a.T4;
        float q = 0.f;
        for (int t = 0; t < a.T4; ++t) { const float d = p[t] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(q / (float)a.T4 + a.eps);
        sstat[(row * a.B + b) * 2] = mean; sstat[(row * a.B + b) * 2 + 1] = rstd;
        float* st = a.stats + ((long long)b * a.Mtot + row_cx(row)) * 2;
        st[0] = mean; st[1] = rstd;
    }
    __syncthreads();
    // ---- affine + activation + residual + store
    const int out_rows = glu ? 8 : 16;
    for (int e = tid; e < out_rows * a.N; e += kTrunkThreads) {
        const int row = e / a.N, nn = e - row * a.N;
        const int c = r0 + row;
        const int b = nn / a.T4, t = nn - b * a.T4;
        const float m0 = sstat[(row * a.B + b) * 2], s0 = sstat[(row * a.B + b) * 2 + 1];
        const float z0 = (tile[row * TLP + nn] - m0) * s0 * a.gamma0[c] + a.beta0[c];
        float y;
        if (glu) {
            const int rg = row + 8;
            const float m1 = sstat[(rg * a.B + b) * 2], s1 = sstat[(rg * a.B + b) * 2 + 1];
            const float z1 = (tile[rg * TLP + nn] - m1) * s1 * a.gamma1[c] + a.beta1[c];
            y = z0 * sigmoidf_(z1);
        } else {
            y = z0;
        }
        const long long yo = (long long)b * a.y_sn + (long long)c * a.y_sc + t;
        if (a.res) y += a.res[yo];
        a.y[yo] = y;
    }
}

// =====================================================================================================================
// Persistent trunk forward: residual blocks + conv1dto2d in one launch (see trunk.h).
// Same arithmetic as trunk_layer_kernel per layer (same MFMA order, same two-pass statistics): results are bit-identical
// to the per-layer launches.  What changes is the hand-off: activations are stored write-through (sc1), every storing wave
// drains its stores, ONE lane bumps the layer's arrival counter, and every workgroup waits for all arrivals with ONE relaxed
// poller + ONE agent-scope acquire before it stages the next layer's input (placement-independent; every spin is bounded).
// =====================================================================================================================
constexpr int kNetWaves = 8;
constexpr int kNetThreads = 64 * kNetWaves;
constexpr int kNetGrid = 64;
static_assert(kNetWaves == kTrunkWaves, "the persistent kernels share trunk_epi_floats with the per-layer kernel");

__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_arrivals(unsigned* ctr, unsigned target)
{
    for (unsigned spins = 0; spins < (1u << 21); ++spins) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// stage X[ci][b][t] (all Cin channels), same layout as trunk_layer_kernel (rows [4 zeros | T4 values] at k = 3, padded channel pitch)
// `fresh`: the source was written by OTHER workgroups of this launch with write-through (sc1) stores: read it with sc1 loads, which
// are served by L2 / memory and never by this CU's (possibly stale) L1 -- no acquire fence needed (Guideline 16, R1 with sc1 on both sides)
typedef int v4i_ __attribute__((ext_vector_type(4)));
template <int KW>
__device__ __forceinline__ void net_stage_x(const float* __restrict__ xsrc, float* Xs, int Cin, int B, int T4, int tid, bool fresh)
{
    constexpr int DOFF = (KW == 3) ? 4 : 0;
    const int TP = T4 + DOFF, RS = trunk_rs(B, T4, KW);
    if (((T4 & 3) == 0) && ((reinterpret_cast<unsigned long long>(xsrc) & 15ull) == 0)) {
        const int nf4 = (Cin * B * T4) >> 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc), 0, nf4 * 16, 0x00020000);
        for (int f0 = 0; f0 < nf4; f0 += 4 * kNetThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u * kNetThreads + tid;
                if (fresh) v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, f * 16, 0, 16));    // out of range -> 0
                else v[u] = (f < nf4) ? *reinterpret_cast<const float4*>(xsrc + 4ll * f) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u * kNetThreads + tid;
                if (f < nf4) {
                    const int e = 4 * f;
                    const int r = e / T4, t = e - r * T4;
                    const int ci = r / B, b = r - ci * B;
                    float* dd = Xs + ci * RS + b * TP + DOFF + t;
                    if constexpr (KW == 3) *reinterpret_cast<float4*>(dd) = v[u];
                    else { *reinterpret_cast<float2*>(dd) = make_float2(v[u].x, v[u].y); *reinterpret_cast<float2*>(dd + 2) = make_float2(v[u].z, v[u].w); }
                }
            }
        }
    } else {
        for (int i = tid; i < Cin * B * T4; i += kNetThreads) {
            const int r = i / T4, t = i - r * T4;
            const int ci = r / B, b = r - ci * B;
            const float* q = xsrc + i;
            Xs[ci * RS + b * TP + DOFF + t] = fresh ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
    }
    trunk_zero_gaps(Xs, Cin, B, TP, RS, DOFF, tid, kNetThreads);
}

// weights of one tile: all of a wave's loads issued at once (the first CH super-groups = everything for the generator's shapes)
// (one register array for both kernel widths: a layer uses [4][3] at k = 3 and the first two columns at k = 1 -- two separate arrays
// were both live across the layer loop and pushed the 48-column kernel into scratch)
template <int KW>
struct NetW { float4 wb[4][3]; };

template <int KW>
__device__ __forceinline__ void net_load_w(const TrunkLayerDesc& d, int tile, NetW<KW>& w)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int K = d.Cin * KW;
    const int glu = (d.mode == TRUNK_IN_GLU);
    const int rows = d.rows, rows_tot = glu ? 2 * rows : rows;
    const int r0 = tile * rows;
    const int k_count = K / kNetWaves;
    const float* arow;
    if (glu) arow = (l15 < rows) ? (d.a0 + (long long)(r0 + l15) * K) : (d.a1 + (long long)(r0 + (l15 < rows_tot ? l15 - rows : 0)) * K);
    else arow = d.a0 + (long long)(r0 + (l15 < rows ? l15 : rows - 1)) * K;       // idle MFMA rows re-read the last row (results unused)
    constexpr int NQ = (KW == 3) ? 3 : 2;
    constexpr int GK = 16 * NQ;
    const int sgroups = k_count / GK;
    const float* ap = arow + wave * k_count + 4 * NQ * kq;       // (lane quarter kq owns whole channels: see the header)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* pp = ap + (long long)(c < sgroups ? c : 0) * GK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) w.wb[c][q] = *reinterpret_cast<const float4*>(pp + 4 * q);
    }
}

// one tile of output rows of one layer: conv (K split over the 8 waves) + bias + IN + GLU / residual, stores included
// Epilogue operands of one tile, requested together with the weights, BEFORE the wait for the previous layer's activations -- bias /
// gamma / beta / the residual input are 1-1.5 us of dependent global-load latency otherwise (measured with an in-kernel clock: the
// epilogue was 5-6 us of a 10 us layer, the arrival wait 0.4 us).  A tile has up to 16 x 64 conv outputs = two per thread (bias[2]) and
// up to 8 x 64 normalised outputs = one per thread.
struct NetE { float bias[2], g0, b0, g1, b1, res; };
__device__ __forceinline__ void net_load_e(const TrunkLayerDesc& d, int B, int T4, int tile, NetE& e)
{
    const int tid = threadIdx.x;
    const int N = B * T4;
    const int glu = (d.mode == TRUNK_IN_GLU);
    const int rows = d.rows, rows_tot = glu ? 2 * rows : rows, r0 = tile * rows;
    e = NetE{{0.f, 0.f}, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int el = tid + it * kNetThreads;
        if (el < rows_tot * N) {
            const int mrow = el / N;
            e.bias[it] = glu ? ((mrow < rows) ? d.bias0[r0 + mrow] : d.bias1[r0 + mrow - rows]) : d.bias0[r0 + mrow];
        }
    }
    if (tid < rows * N) {
        const int row = tid / N, col = tid - row * N, c = r0 + row;
        e.g0 = d.gamma0[c]; e.b0 = d.beta0[c];
        if (glu) { e.g1 = d.gamma1[c]; e.b1 = d.beta1[c]; }
        if (d.res) {
            const int b = col / T4, t = col - b * T4;
            e.res = __hip_atomic_load(d.res + (long long)b * d.y_sn + (long long)c * d.y_sc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (this workgroup's own rows, two layers back)
        }
    }
}

template <int KW, int NA>
__device__ __forceinline__ void net_tile(const TrunkLayerDesc& d, int B, int T4, float eps, int tile, const float* Xs, float* epi, bool wt,
                                         NetW<KW>& pre, const NetE& pe)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int DOFF = (KW == 3) ? 4 : 0;
    const int TP = T4 + DOFF, RS = trunk_rs(B, T4, KW);
    const int N = B * T4, K = d.Cin * KW;
    const int TLP = N;                             // (tile rows N apart: thread el reads / writes word el -- no bank conflicts; the statistics are shuffles)
    const int glu = (d.mode == TRUNK_IN_GLU);
    const int rows = d.rows;                       // per branch
    const int rows_tot = glu ? 2 * rows : rows;    // MFMA rows in use (<= 16)
    const int r0 = tile * rows;
    const int Mtot = glu ? 2 * d.M : d.M;
    const int k_count = K / kNetWaves;
    const int k_begin = wave * k_count;
    const float* arow;
    {
        const int i = l15;
        if (glu) arow = (i < rows) ? (d.a0 + (long long)(r0 + i) * K) : (d.a1 + (long long)(r0 + (i < rows_tot ? i - rows : 0)) * K);
        else arow = d.a0 + (long long)(r0 + (i < rows ? i : rows - 1)) * K;       // idle MFMA rows re-read the last row (results unused)
    }
    constexpr int NQ = (KW == 3) ? 3 : 2;
    constexpr int GK = 16 * NQ;
    constexpr int CH = 4;
    const int sgroups = k_count / GK;
    const float* ap = arow + k_begin + 4 * NQ * kq;
    float4 (&wb)[CH][3] = pre.wb;                  // super-groups 0..CH-1 were requested by net_load_w (before the layer's wait)
    int off[NA][NQ][4];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        const int n = l15 + 16 * h;
        int xcol = B * TP, live = 0;
        if (n < N) { const int b = n / T4, t = n - b * T4; xcol = b * TP + DOFF - (KW - 1) / 2 + t; live = 1; }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = 4 * (NQ * kq + q) + j;
                off[h][q][j] = (kr / KW) * RS + xcol + (live ? (kr % KW) : 0);
            }
    }
    f32x4 acc[NA];
#pragma unroll
    for (int h = 0; h < NA; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* xs = Xs + (k_begin / KW) * RS;
        for (int sg0 = 0; sg0 < sgroups; sg0 += CH) {
            if (sg0 > 0) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float* pp = ap + (long long)(sg0 + c < sgroups ? sg0 + c : 0) * GK;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) wb[c][q] = *reinterpret_cast<const float4*>(pp + 4 * q);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (sg0 + c < sgroups) {
                    float xv[NQ][4][NA];          // all operand reads of the super-group first: one LDS latency instead of one per MFMA
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int h = 0; h < NA; ++h) xv[q][j][h] = xs[off[h][q][j]];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float av[4] = {wb[c][q].x, wb[c][q].y, wb[c][q].z, wb[c][q].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int h = 0; h < NA; ++h) acc[h] = MFMA16(av[j], xv[q][j][h], acc[h]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, NQ * 4 * NA, 0);       // (left alone the scheduler sinks every read to its MFMA)
                    __builtin_amdgcn_sched_group_barrier(0x008, NQ * 4 * NA, 0);
                    xs += (GK / KW) * RS;
                }
            }
        }
    }
    // ---- cross-wave K reduction: red[wave][h][reg][lane]
    float* red = epi;
    float* tl = epi + kNetWaves * NA * 4 * kRedPitch;     // [16][TLP]
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NA + h) * 4 + r) * kRedPitch + lane] = acc[h][r];
    __syncthreads();
    // one thread per MFMA output element (mrow, col) (two at more than 32 columns), a row's N columns in consecutive threads: sum of the
    // 8 waves' partials + bias
    auto row_cx = [&](int row) { return glu ? ((row < rows) ? (r0 + row) : (d.M + r0 + row - rows)) : (r0 + row); };
#pragma unroll
    for (int it = 0; it < (NA > 2 ? 2 : 1); ++it) {
        const int el = tid + it * kNetThreads;
        if (el < rows_tot * N) {
            const int mrow = el / N, col = el - mrow * N;
            const int h = col >> 4, ln = ((mrow >> 2) << 4) + (col & 15), r = mrow & 3;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kNetWaves; ++w) v += red[((w * NA + h) * 4 + r) * kRedPitch + ln];
            v += pe.bias[it];
            tl[mrow * TLP + col] = v;
            d.conv_out[(long long)row_cx(mrow) * N + col] = v;
        }
    }
    __syncthreads();
    // statistics of (row, sample) over its T4 columns, computed by every thread of the segment (LDS broadcast reads) instead of by
    // one thread per row behind two more barriers
    if (tid < rows * N) {
        const int row = tid / N, col = tid - row * N;
        const int c = r0 + row;
        const int b = col / T4, t = col - b * T4;
        const bool pow2 = (T4 & (T4 - 1)) == 0;            // (uniform) the T4 lanes of a (row, sample) segment are consecutive and aligned
        auto stats_of = [&](int mrow, float& mean, float& rstd) {
            const float* p = tl + mrow * TLP + b * T4;
            float sum, q;
            if (pow2) {                                   // butterfly over the segment's lanes: 2 log2(T4) shuffles instead of 2 T4 LDS reads
                sum = p[t];
                for (int m = T4 >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
                mean = sum / (float)T4;
                const float dd = p[t] - mean;
                q = dd * dd;
                for (int m = T4 >> 1; m >= 1; m >>= 1) q += __shfl_xor(q, m);
            } else {
                sum = 0.f;
                for (int u = 0; u < T4; ++u) sum += p[u];
                mean = sum / (float)T4;
                q = 0.f;
                for (int u = 0; u < T4; ++u) { const float dd = p[u] - mean; q += dd * dd; }
            }
            rstd = 1.0f / sqrtf(q / (float)T4 + eps);
            if (t == 0) {
                float* st = d.stats + ((long long)b * Mtot + row_cx(mrow)) * 2;
                st[0] = mean; st[1] = rstd;
            }
        };
        float m0, s0;
        stats_of(row, m0, s0);
        const float z0 = (tl[row * TLP + col] - m0) * s0 * pe.g0 + pe.b0;
        float y;
        if (glu) {
            float m1, s1;
            stats_of(row + rows, m1, s1);
            const float z1 = (tl[(row + rows) * TLP + col] - m1) * s1 * pe.g1 + pe.b1;
            y = z0 * sigmoidf_(z1);
        } else {
            y = z0;
        }
        const long long yo = (long long)b * d.y_sn + (long long)c * d.y_sc + t;
        if (d.res) y += pe.res;
        if (wt) st_wt(d.y + yo, y); else d.y[yo] = y;
    }
}

template <int NA>
__global__ void __launch_bounds__(kNetThreads) trunk_fwd_net_kernel(const Twin<TrunkFwdNetArgs> tw)
{
    const TrunkFwdNetArgs& a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* epi = smem + a.x_floats;
    float* flag = epi + trunk_epi_floats(NA) - 16;
    const int tid = threadIdx.x;
    for (int l = 0; l < a.nlayers; ++l) {
        const TrunkLayerDesc d = a.L[l];          // (by value: the fields are loaded once per layer, not inside the loops below)
        const int ntiles = d.M / d.rows;
        // this layer's weights do not depend on the previous layer: request them BEFORE waiting for its activations
        NetW<3> w3; NetE pe;
        NetW<1>& w1 = reinterpret_cast<NetW<1>&>(w3);
        const bool mine = (int)blockIdx.x < ntiles;
        if (mine) { if (d.KW == 3) net_load_w<3>(d, blockIdx.x, w3); else net_load_w<1>(d, blockIdx.x, w1); net_load_e(d, a.B, a.T4, blockIdx.x, pe); }
        if (l > 0) {
            if (tid == 0) {
                const bool ok = wait_arrivals(a.sync + (l - 1), gridDim.x);
                if (!ok) __hip_atomic_store(a.sync + MCVC_TRUNK_SYNC_WORDS - 1, 1u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag[0] = ok ? 1.f : 0.f;
            }
            // barrier WITHOUT the vmcnt(0) that __syncthreads() implies: the weight / operand loads requested above stay in flight while
            // the activations are staged (their latencies overlap instead of adding up)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (flag[0] == 0.f) {                  // (uniform) a workgroup never arrived: give up instead of hanging the device ...
                // ... and make the failure VISIBLE: the pass's result (the last layer's output rows this workgroup owns) becomes NaN, which
                // reaches every loss term of the step; the error word (read by mcvc_gen_trunk_fault) names the layer.  A missing arrival
                // stalls every workgroup at the same layer, so all of them take this path and the whole output is poisoned.
                const TrunkLayerDesc& dl = a.L[a.nlayers - 1];
                const int N = a.B * a.T4;
                for (int tile = blockIdx.x; tile < dl.M / dl.rows; tile += gridDim.x)
                    for (int i = tid; i < dl.rows * N; i += kNetThreads) {
                        const int row = i / N, col = i - row * N, b = col / a.T4, t = col - b * a.T4;
                        dl.y[(long long)b * dl.y_sn + (long long)(tile * dl.rows + row) * dl.y_sc + t] = __builtin_nanf("");
                    }
                return;
            }
        }
        if (d.KW == 3) net_stage_x<3>(d.x, Xs, d.Cin, a.B, a.T4, tid, l > 0); else net_stage_x<1>(d.x, Xs, d.Cin, a.B, a.T4, tid, l > 0);
        __syncthreads();
        const bool wt = (l + 1 < a.nlayers);       // the last layer's output is consumed after the kernel boundary
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            if (tile != (int)blockIdx.x) { if (d.KW == 3) net_load_w<3>(d, tile, w3); else net_load_w<1>(d, tile, w1); net_load_e(d, a.B, a.T4, tile, pe); }
            if (d.KW == 3) net_tile<3, NA>(d, a.B, a.T4, a.eps, tile, Xs, epi, wt, w3, pe);
            else net_tile<1, NA>(d, a.B, a.T4, a.eps, tile, Xs, epi, wt, w1, pe);
        }
        if (wt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // EVERY storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0 && !(a.fault_inject && l == 0 && blockIdx.x == 0))      // (test hook: workgroup 0 "never arrives" at layer 0)
                __hip_atomic_fetch_add(a.sync + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// =====================================================================================================================
// Persistent trunk backward (see trunk.h)
// =====================================================================================================================
// stage X' = InstanceNorm (+ gated GLU) backward of dy for ALL Cx channels into Xs (zero halo, k = 3), same arithmetic and the same
// 4-lanes-per-row decomposition as the PRE path of trunk_layer_kernel; owner workgroups store X' and add d(gamma), d(beta)
__device__ __forceinline__ void bnet_stage_pre(const TrunkBwdNetArgs& a, const TrunkBwdLayerDesc& d, float* Xs, float* rsum, int B, int T4, int tid)
{
    constexpr int DOFF = 4;
    const int TP = T4 + DOFF, RS = trunk_rs(B, T4, 3);
    const int C = d.C;
    const bool pglu = (d.pre == 2);
    const int Cx = pglu ? 2 * C : C;
    const int PB = d.pxB > 0 ? d.pxB : B;                  // samples per channel of the forward pass's tensor (prefix backward: > B)
    const float invT = 1.0f / (float)T4;
    const int E = (T4 + 3) >> 2;
    const int items = Cx * B * 4;
    const int nwg = (int)gridDim.x, me = (int)blockIdx.x;
    if (T4 == 16 && !(d.flags & TBWD_SLAB_DY) && ((reinterpret_cast<unsigned long long>(d.dy) | reinterpret_cast<unsigned long long>(d.px) |
                                                   reinterpret_cast<unsigned long long>(d.xout)) & 15ull) == 0) {
        // the trainer's shape (64 frames): a lane's four elements are ONE 16-byte load per tensor, and the loads of four items per thread
        // are all in flight before the first is used -- every workgroup stages every channel, so this loop IS the layer's latency
        constexpr int U = 4;
        const bool fresh = (d.flags & TBWD_DY_FRESH) != 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy), 0, C * B * 16 * 4, 0x00020000);
        for (int it0 = 0; it0 < items; it0 += U * kNetThreads) {
            float4 vd[U], v0[U], v1[U];
            float g0[U], b0[U], g1[U], b1[U], m0[U], r0[U], m1[U], r1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = it0 + u * kNetThreads + tid;
                const bool live = item < items;
                const int q4 = item & 3, row = live ? (item >> 2) : 0;
                const int cx = row / B, b = row - cx * B;
                const bool gate = pglu && cx >= C;
                const int c = gate ? cx - C : cx;
                const long long ro = ((long long)c * B + b) * 16 + q4 * 4;
                const long long po = ((long long)c * PB + b) * 16 + q4 * 4;
                vd[u] = fresh ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro * 4), 0, 16))
                              : *reinterpret_cast<const float4*>(d.dy + ro);
                v0[u] = *reinterpret_cast<const float4*>(d.px + po);
                v1[u] = pglu ? *reinterpret_cast<const float4*>(d.px + po + (long long)C * PB * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                g0[u] = d.g0[c]; b0[u] = d.b0[c];
                g1[u] = pglu ? d.g1[c] : 0.f; b1[u] = pglu ? d.b1[c] : 0.f;
                const float* st = d.stats + (long long)b * Cx * 2;
                m0[u] = st[2 * c]; r0[u] = st[2 * c + 1];
                m1[u] = pglu ? st[2 * (c + C)] : 0.f; r1[u] = pglu ? st[2 * (c + C) + 1] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = it0 + u * kNetThreads + tid;
                const bool live = item < items;
                const int q4 = item & 3, row = live ? (item >> 2) : 0;
                const int cx = row / B, b = row - cx * B;
                const bool gate = pglu && cx >= C;
                const float dv[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
                const float x0[4] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w};
                const float x1[4] = {v1[u].x, v1[u].y, v1[u].z, v1[u].w};
                float dzv[4], xhv[4];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh0 = (x0[e] - m0[u]) * r0[u];
                    float dz, xh;
                    if (pglu) {
                        const float xh1 = (x1[e] - m1[u]) * r1[u];
                        const float sg = sigmoidf_(xh1 * g1[u] + b1[u]);
                        if (gate) { dz = dv[e] * (xh0 * g0[u] + b0[u]) * sg * (1.0f - sg); xh = xh1; }
                        else { dz = dv[e] * sg; xh = xh0; }
                    } else { dz = dv[e]; xh = xh0; }
                    dzv[e] = live ? dz : 0.f; xhv[e] = live ? xh : 0.f;
                    s1 += dzv[e]; s2 += dzv[e] * xhv[e];
                }
                s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
                s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
                if (live) {
                    const bool mine = (cx % nwg) == me;
                    const float gr = gate ? g1[u] * r1[u] : g0[u] * r0[u];
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gr * (dzv[e] - s1 * invT - xhv[e] * (s2 * invT));
                    *reinterpret_cast<float4*>(Xs + cx * RS + b * TP + DOFF + q4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
                    if (mine) *reinterpret_cast<float4*>(d.xout + ((long long)cx * B + b) * 16 + q4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
                    if (q4 == 0 && mine) { rsum[((cx / nwg) * B + b) * 2] = s1; rsum[((cx / nwg) * B + b) * 2 + 1] = s2; }
                }
            }
        }
    } else
    for (int it0 = 0; it0 < items; it0 += kNetThreads) {
        const int item = it0 + tid;
        const bool live = item < items;
        const int q4 = item & 3, row = live ? (item >> 2) : 0;
        const int cx = row / B, b = row - cx * B;
        const bool gate = pglu && cx >= C;
        const int c = gate ? cx - C : cx;
        const float g0 = d.g0[c], b0 = d.b0[c];
        const float g1 = pglu ? d.g1[c] : 0.f, b1 = pglu ? d.b1[c] : 0.f;
        const float* st = d.stats + (long long)b * Cx * 2;
        const float m0 = st[2 * c], r0 = st[2 * c + 1];
        const float m1 = pglu ? st[2 * (c + C)] : 0.f, r1 = pglu ? st[2 * (c + C) + 1] : 1.f;
        const int t0 = q4 * E;
        const float* dyr = d.dy + ((long long)c * B + b) * T4 + t0;
        const float* x0r = d.px + ((long long)c * PB + b) * T4 + t0;
        const float* x1r = d.px + ((long long)(c + C) * PB + b) * T4 + t0;
        float vd[8], v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = live && e < E && (t0 + e) < T4;
            // dy written by other workgroups of this launch (write-through): sc1 loads, never this CU's possibly stale L1
            vd[e] = ok ? ((d.flags & TBWD_DY_FRESH) ? __hip_atomic_load(dyr + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dyr[e]) : 0.f;
            if (ok && (d.flags & TBWD_SLAB_DY))
                for (int sl = 1; sl < a.nslab; ++sl) vd[e] += a.slabs[(long long)(sl - 1) * a.slab_stride + ((long long)c * B + b) * T4 + t0 + e];
            v0[e] = ok ? x0r[e] : 0.f; v1[e] = (ok && pglu) ? x1r[e] : 0.f;
        }
        float dzv[8], xhv[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = e < E && (t0 + e) < T4;
            const float xh0 = (v0[e] - m0) * r0;
            float dz, xh;
            if (pglu) {
                const float xh1 = (v1[e] - m1) * r1;
                const float sg = sigmoidf_(xh1 * g1 + b1);
                if (gate) { dz = vd[e] * (xh0 * g0 + b0) * sg * (1.0f - sg); xh = xh1; }
                else { dz = vd[e] * sg; xh = xh0; }
            } else { dz = vd[e]; xh = xh0; }
            dzv[e] = ok ? dz : 0.f; xhv[e] = ok ? xh : 0.f;
            s1 += dzv[e]; s2 += dzv[e] * xhv[e];
        }
        s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
        s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
        if (live) {
            const bool mine = (cx % nwg) == me;
            const float gr = gate ? g1 * r1 : g0 * r0;
            float* xrow = Xs + cx * RS + b * TP + DOFF;
            float* od = mine ? (d.xout + ((long long)cx * B + b) * T4 + t0) : nullptr;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < E && (t0 + e) < T4) {
                    const float dxv = gr * (dzv[e] - s1 * invT - xhv[e] * (s2 * invT));
                    xrow[t0 + e] = dxv;
                    if (od) od[e] = dxv;
                }
            }
            if (q4 == 0 && mine) { rsum[((cx / nwg) * B + b) * 2] = s1; rsum[((cx / nwg) * B + b) * 2 + 1] = s2; }
        }
    }
    trunk_zero_gaps(Xs, Cx, B, TP, RS, DOFF, tid, kNetThreads);
    __syncthreads();
    // d(gamma), d(beta) of the owned channels: sums over the samples in a fixed order
    for (int j = tid; j * nwg + me < Cx; j += kNetThreads) {
        const int cx = j * nwg + me;
        const bool gate = pglu && cx >= C;
        const int c = gate ? cx - C : cx;
        float dgam = 0.f, dbet = 0.f;
        for (int b = 0; b < B; ++b) { dbet += rsum[(j * B + b) * 2]; dgam += rsum[(j * B + b) * 2 + 1]; }
        float* dg = gate ? d.dg1 : d.dg0;
        float* db = gate ? d.db1 : d.db0;
        if (dg) dg[c] += dgam;
        if (db) db[c] += dbet;
    }
}

// one tile of `rows` output rows: transposed conv over the staged X' (K split over the 8 waves), summed in a fixed order, stored
// write-through (+ the old value when the layer accumulates onto the skip connection's gradient)
template <int NA>
__device__ __forceinline__ void bnet_tile(const TrunkBwdLayerDesc& d, int B, int T4, int tile, const float* Xs, float* epi, bool wt,
                                          NetW<3>& pre, float old)
{
    constexpr int KW = 3, DOFF = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int TP = T4 + DOFF, RS = trunk_rs(B, T4, 3);
    const int N = B * T4;
    const int Cx = (d.pre == 2) ? 2 * d.C : d.C;
    const int K = Cx * KW;
    const int rows = d.rows, r0 = tile * rows;
    const int k_count = K / kNetWaves;
    const int k_begin = wave * k_count;
    const float* arow = d.wt + (long long)(r0 + (l15 < rows ? l15 : rows - 1)) * K;
    constexpr int NQ = 3, GK = 48, CH = 4;
    const int sgroups = k_count / GK;
    const float* ap = arow + k_begin + 4 * NQ * kq;
    float4 (&wb)[CH][3] = pre.wb;
    int off[NA][NQ][4];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        const int n = l15 + 16 * h;
        int xcol = B * TP, live = 0;
        if (n < N) { const int b = n / T4, t = n - b * T4; xcol = b * TP + DOFF - 1 + t; live = 1; }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = 4 * (NQ * kq + q) + j;
                off[h][q][j] = (kr / KW) * RS + xcol + (live ? (kr % KW) : 0);
            }
    }
    f32x4 acc[NA];
#pragma unroll
    for (int h = 0; h < NA; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* xs = Xs + (k_begin / KW) * RS;
        for (int sg0 = 0; sg0 < sgroups; sg0 += CH) {
            if (sg0 > 0) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float* pp = ap + (long long)(sg0 + c < sgroups ? sg0 + c : 0) * GK;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) wb[c][q] = *reinterpret_cast<const float4*>(pp + 4 * q);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (sg0 + c < sgroups) {
                    float xv[NQ][4][NA];
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int h = 0; h < NA; ++h) xv[q][j][h] = xs[off[h][q][j]];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float av[4] = {wb[c][q].x, wb[c][q].y, wb[c][q].z, wb[c][q].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int h = 0; h < NA; ++h) acc[h] = MFMA16(av[j], xv[q][j][h], acc[h]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, NQ * 4 * NA, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NQ * 4 * NA, 0);
                    xs += (GK / KW) * RS;
                }
            }
        }
    }
    float* red = epi;
#pragma unroll
    for (int h = 0; h < NA; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NA + h) * 4 + r) * kRedPitch + lane] = acc[h][r];
    __syncthreads();
    if (tid < rows * N) {
        const int mrow = tid / N, col = tid - mrow * N;
        const int h = col >> 4, ln = ((mrow >> 2) << 4) + (col & 15), r = mrow & 3;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kNetWaves; ++w) v += red[((w * NA + h) * 4 + r) * kRedPitch + ln];
        if (d.flags & TBWD_ACCUMULATE) v += old;
        float* dst = d.out + (long long)(r0 + mrow) * N + col;
        if (wt) st_wt(dst, v); else *dst = v;
    }
}

static_assert(sizeof(Twin<TrunkBwdNetArgs>) <= 4096 && sizeof(Twin<TrunkFwdNetArgs>) <= 4096, "kernel arguments are limited to 4 KB");
template <int NA>
__global__ void __launch_bounds__(kNetThreads) trunk_bwd_net_kernel(const Twin<TrunkBwdNetArgs> tw)
{
    const TrunkBwdNetArgs& a = tw.v[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* rsum = smem + a.x_floats;                       // [ceil(Cx / workgroups)][B][2]
    float* epi = rsum + 16 * 8 * 2 * 2;                    // (Cx <= 1024, 64 workgroups, B <= 8)
    float* flag = epi + kNetWaves * NA * 4 * kRedPitch;
    const int tid = threadIdx.x;
    const int N = a.B * a.T4;
    for (int l = 0; l < a.nlayers; ++l) {
        const TrunkBwdLayerDesc d = a.L[l];       // (by value: loaded once per layer)
        const int ntiles = d.M / d.rows;
        const int Cx = (d.pre == 2) ? 2 * d.C : d.C;
        // this layer's weights (and the value it accumulates onto: this workgroup's own rows) do not depend on the previous layer
        NetW<3> w3;
        TrunkLayerDesc wd{};
        wd.a0 = d.wt; wd.Cin = Cx; wd.KW = 3; wd.M = d.M; wd.mode = TRUNK_PLAIN; wd.rows = d.rows;
        const bool mine = (int)blockIdx.x < ntiles;
        float old = 0.f;
        auto load_old = [&](int tile) {
            float v = 0.f;
            if ((d.flags & TBWD_ACCUMULATE) && tid < d.rows * N) {
                const long long idx = (long long)(tile * d.rows) * N + tid;
                v = __hip_atomic_load(d.out + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d.flags & TBWD_SLAB_OUT)
                    for (int sl = 1; sl < a.nslab; ++sl) v += a.slabs[(long long)(sl - 1) * a.slab_stride + idx];
            }
            return v;
        };
        if (mine) {
            net_load_w<3>(wd, blockIdx.x, w3);
            old = load_old(blockIdx.x);
        }
        if (l > 0) {
            if (tid == 0) {
                const bool ok = wait_arrivals(a.sync + (l - 1), gridDim.x);
                if (!ok) __hip_atomic_store(a.err, 0x100u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag[0] = ok ? 1.f : 0.f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (flag[0] == 0.f) {                  // a workgroup never arrived: poison the result (the trunk's input gradient) and give up
                const TrunkBwdLayerDesc& dl = a.L[a.nlayers - 1];
                for (int tile = blockIdx.x; tile < dl.M / dl.rows; tile += gridDim.x)
                    for (int i = tid; i < dl.rows * N; i += kNetThreads) dl.out[(long long)(tile * dl.rows) * N + i] = __builtin_nanf("");
                return;
            }
        }
        bnet_stage_pre(a, d, Xs, rsum, a.B, a.T4, tid);
        __syncthreads();
        const bool wt = (l + 1 < a.nlayers);
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            if (tile != (int)blockIdx.x) {
                net_load_w<3>(wd, tile, w3);
                old = load_old(tile);
            }
            bnet_tile<NA>(d, a.B, a.T4, tile, Xs, epi, wt, w3, old);
        }
        if (wt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !(a.fault_inject && l == 0 && blockIdx.x == 0))
                __hip_atomic_fetch_add(a.sync + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct ZeroWordsKArgs { unsigned* p; int n; };
__global__ void zero_words_kernel(const Twin<ZeroWordsKArgs> tw)
{
    const ZeroWordsKArgs a = tw.v[blockIdx.z];
    if ((int)threadIdx.x < a.n) a.p[threadIdx.x] = 0u;
}

// dst[c][0..per_row) = bias ? bias[c] : 0   (initial value of an atomically accumulated K-split trunk layer)
struct FillRowsKArgs { float* dst; const float* bias; int C; int per_row; };
__global__ void __launch_bounds__(256) fill_rows_kernel(const Twin<FillRowsKArgs> tw)
{
    const FillRowsKArgs ka_ = tw.v[blockIdx.z];
    float* __restrict__ dst = ka_.dst;
    const float* __restrict__ bias = ka_.bias;
    int C = ka_.C;
    int per_row = ka_.per_row;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * per_row) return;
    dst[i] = bias ? bias[i / per_row] : 0.f;
}

}  // namespace

int mcvc_fill_rows_launch(float* dst, const float* bias, int C, int per_row, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 4.0 * C * per_row);
    mcvc_launch(fill_rows_kernel, dim3((unsigned)cdiv_i(C * per_row, 256)), dim3(256), 0, s, FillRowsKArgs{dst, bias, C, per_row});
    return (int)hipGetLastError();
}

bool mcvc_trunk_applies(int Cin, int KW, int M, int B, int T4, int mode, int ksplit)
{
    if (KW != 1 && KW != 3) return false;
    if (B * T4 > 64 || T4 > 32 || B > 8) return false;
    const int K = Cin * KW;
    if (ksplit < 1 || (K % ksplit) != 0) return false;
    const int kblk = K / ksplit;
    const int gk = (KW == 3) ? 48 : 32;                          // one weight super-group per wave (see the kernel)
    if ((kblk % (gk * kTrunkWaves)) != 0) return false;
    const int rows = (mode == TRUNK_IN_GLU) ? 8 : 16;
    if (M % rows != 0) return false;
    const long long lds = mcvc_trunk_lds_floats(Cin, KW, B, T4, ksplit);
    return lds * 4 <= 150 * 1024;
}

long long mcvc_trunk_lds_floats(int Cin, int KW, int B, int T4, int ksplit)
{
    const long long xs = (long long)(Cin / ksplit) * trunk_rs(B, T4, KW) + 4;
    const long long epi = trunk_epi_floats(trunk_na(B * T4));
    return xs > epi ? xs : epi;
}

template <int KW, int NA, bool PRE>
static int trunk_launch_p(const TrunkArgs& a, dim3 grid, size_t lds, hipStream_t s)
{
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(trunk_layer_kernel<KW, NA, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    mcvc_launch((trunk_layer_kernel<KW, NA, PRE>), grid, dim3(kTrunkThreads), lds, s, a);
    return (int)hipGetLastError();
}

template <int KW, int NA>
static int trunk_launch_t(const TrunkArgs& a, dim3 grid, size_t lds, hipStream_t s)
{
    return a.pre ? trunk_launch_p<KW, NA, true>(a, grid, lds, s) : trunk_launch_p<KW, NA, false>(a, grid, lds, s);
}

template <int KW>
static int trunk_launch_k(const TrunkArgs& a, dim3 grid, size_t lds, hipStream_t s)
{
    switch (trunk_na(a.N)) {
    case 1: return trunk_launch_t<KW, 1>(a, grid, lds, s);
    case 2: return trunk_launch_t<KW, 2>(a, grid, lds, s);
    case 3: return trunk_launch_t<KW, 3>(a, grid, lds, s);
    default: return trunk_launch_t<KW, 4>(a, grid, lds, s);
    }
}

int mcvc_trunk_launch(const TrunkArgs& a, int ksplit, hipStream_t s)
{
    if (!mcvc_trunk_applies(a.Cin, a.KW, a.M, a.B, a.T4, a.mode, ksplit)) return MCVC_ERR_INVALID;
    if (ksplit > 1 && !(a.mode == TRUNK_PLAIN && (a.accumulate || a.slabs))) return MCVC_ERR_INVALID;
    if (a.slab_all && !(a.mode == TRUNK_PLAIN && a.slabs)) return MCVC_ERR_INVALID;
    if (a.pre && (a.mode != TRUNK_PLAIN || a.T4 > 32 || !a.pre_x || !a.pre_stats || !a.pre_out || (a.pre_xB > 0 && a.pre_xB < a.B))) return MCVC_ERR_INVALID;
    const int rows = (a.mode == TRUNK_IN_GLU) ? 8 : 16;
    dim3 grid((unsigned)(a.M / rows), (unsigned)ksplit);
    size_t lds = (size_t)mcvc_trunk_lds_floats(a.Cin, a.KW, a.B, a.T4, ksplit) * sizeof(float);
    if (a.pre) {          // + [channels of a K slice][B][2] row sums behind the staged slice
        const long long xs = (long long)(a.Cin / ksplit) * trunk_rs(a.B, a.T4, a.KW) + 4 + (long long)(a.Cin / ksplit) * a.B * 2;
        if ((size_t)xs * sizeof(float) > lds) lds = (size_t)xs * sizeof(float);
        if (lds > 160 * 1024) return MCVC_ERR_INVALID;
    }
    const double mt = (a.mode == TRUNK_IN_GLU) ? 2.0 * a.M : (double)a.M;
    TraceScope ts(K_TRUNK, s, 2.0 * mt * a.K * a.N, 4.0 * (mt * a.K + (double)a.Cin * a.N + 3.0 * mt * a.N));
    return a.KW == 3 ? trunk_launch_k<3>(a, grid, lds, s) : trunk_launch_k<1>(a, grid, lds, s);
}

// widest staged input of the persistent forward: 512 channels, k = 3
static long long net_fwd_lds_floats(int B, int T4) { return (long long)512 * trunk_rs(B, T4, 3) + 4 + trunk_epi_floats(trunk_na(B * T4)); }
// ... of the persistent backward: 1024 channels (value | gate), k = 3, + the owners' row sums
static long long net_bwd_lds_floats(int B, int T4) { return (long long)1024 * trunk_rs(B, T4, 3) + 4 + 16 * 8 * 2 * 2 + kNetWaves * trunk_na(B * T4) * 4 * kRedPitch + 16; }

// The persistent kernels' workgroups wait for each other inside the kernel: all kNetGrid of them (x 2 in a grouped launch) must be
// resident at once, each on a compute unit of its own when its LDS exceeds half a CU's.  The caller states how many such passes it keeps in
// flight (mcvc_set_trunk_residency); the kernels are used only when the device has the CUs for all of them.
static int g_trunk_passes_in_flight = 2;          // grouped launches of 2 x 64 workgroups
static int device_cus()
{
    static const int n = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0; return v; }();
    return n;
}
int mcvc_trunk_set_passes_in_flight(int n) { const int was = g_trunk_passes_in_flight; if (n >= 1) g_trunk_passes_in_flight = n; return was; }
static bool trunk_residency_ok() { const int cus = device_cus(); return cus <= 0 || kNetGrid * g_trunk_passes_in_flight <= cus; }

bool mcvc_trunk_net_applies(int B, int T4)
{
    if (B < 1 || B > 8 || T4 < 1 || T4 > 32 || B * T4 > 48) return false;          // (64 columns never fit the LDS beside 512 staged channels)
    return trunk_residency_ok() && net_fwd_lds_floats(B, T4) * 4 <= 160 * 1024;
}

bool mcvc_trunk_bwd_net_applies(int B, int T4)
{
    if (B < 1 || B > 8 || T4 < 4 || T4 > 32 || B * T4 > 32) return false;
    return trunk_residency_ok() && net_bwd_lds_floats(B, T4) * 4 <= 160 * 1024;
}

// test hook (mcvc_debug_trunk_fault_inject): the next persistent launches lose one arrival, so that the give-up path can be exercised
static int g_trunk_fault_inject = 0;
int mcvc_trunk_set_fault_inject(int on) { const int was = g_trunk_fault_inject; g_trunk_fault_inject = on ? 1 : 0; return was; }

int mcvc_trunk_fwd_net_launch(TrunkFwdNetArgs& a, hipStream_t s)
{
    if (!mcvc_trunk_net_applies(a.B, a.T4) || a.nlayers < 1 || a.nlayers > MCVC_TRUNK_NET_LAYERS || !a.sync) return MCVC_ERR_INVALID;
    long long xmax = 0;
    double flops = 0.0, bytes = 0.0;
    const int N = a.B * a.T4;
    for (int l = 0; l < a.nlayers; ++l) {
        const TrunkLayerDesc& d = a.L[l];
        if (d.KW != 1 && d.KW != 3) return MCVC_ERR_INVALID;
        const int K = d.Cin * d.KW, gk = (d.KW == 3) ? 48 : 32;
        if (K % (gk * kNetWaves) != 0 || d.rows < 1 || d.rows > 16 || d.M % d.rows != 0) return MCVC_ERR_INVALID;
        if (d.mode == TRUNK_IN_GLU && d.rows > 8) return MCVC_ERR_INVALID;
        if (d.mode != TRUNK_IN_GLU && d.mode != TRUNK_IN) return MCVC_ERR_INVALID;
        const int rows_tot = d.mode == TRUNK_IN_GLU ? 2 * d.rows : d.rows;
        if (d.rows * N > kNetThreads || rows_tot * N > 2 * kNetThreads) return MCVC_ERR_INVALID;       // epilogue: one / two elements per thread
        const long long x = (long long)d.Cin * trunk_rs(a.B, a.T4, d.KW) + 4;
        if (x > xmax) xmax = x;
        const double mt = (d.mode == TRUNK_IN_GLU) ? 2.0 * d.M : (double)d.M;
        flops += 2.0 * mt * K * N;
        bytes += 4.0 * (mt * K + (double)d.Cin * N + 3.0 * mt * N);
    }
    a.x_floats = (int)((xmax + 3) & ~3LL);
    a.fault_inject = g_trunk_fault_inject;
    const int NA = trunk_na(N);
    const size_t lds = ((size_t)a.x_floats + trunk_epi_floats(NA)) * sizeof(float);
    if (lds > 160 * 1024) return MCVC_ERR_INVALID;
    hipError_t e = hipSuccess;
    // arrival counters back to zero (a kernel, not a memset: it takes part in grouped launches); the error word is sticky (mcvc_gen_trunk_fault)
    mcvc_launch(zero_words_kernel, dim3(1), dim3(64), 0, s, ZeroWordsKArgs{a.sync, MCVC_TRUNK_SYNC_WORDS - 1});
    TraceScope ts(K_TRUNK, s, flops, bytes);
    static bool done[4] = {false, false, false, false};
    typedef void (*KernT)(const Twin<TrunkFwdNetArgs>);
    const KernT fns[4] = {nullptr, trunk_fwd_net_kernel<1>, trunk_fwd_net_kernel<2>, trunk_fwd_net_kernel<3>};
    if (!done[NA]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[NA]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done[NA] = true;
    }
    mcvc_launch(fns[NA], dim3(kNetGrid), dim3(kNetThreads), lds, s, a);
    return (int)hipGetLastError();
}

int mcvc_trunk_bwd_net_launch(TrunkBwdNetArgs& a, hipStream_t s)
{
    if (!mcvc_trunk_bwd_net_applies(a.B, a.T4) || a.nlayers < 1 || a.nlayers > MCVC_TRUNK_BWD_LAYERS || !a.sync || !a.err) return MCVC_ERR_INVALID;
    long long xmax = 0;
    double flops = 0.0, bytes = 0.0;
    const int N = a.B * a.T4;
    for (int l = 0; l < a.nlayers; ++l) {
        const TrunkBwdLayerDesc& d = a.L[l];
        const int Cx = (d.pre == 2) ? 2 * d.C : d.C;
        const int K = Cx * 3;
        if ((d.pre != 1 && d.pre != 2) || Cx > 1024 || K % (48 * kNetWaves) != 0 || d.rows < 1 || d.rows > 16 || d.M % d.rows != 0 || d.rows * N > kNetThreads ||
            (d.pxB > 0 && d.pxB < a.B))
            return MCVC_ERR_INVALID;
        const long long x = (long long)Cx * trunk_rs(a.B, a.T4, 3) + 4;
        if (x > xmax) xmax = x;
        flops += 2.0 * d.M * K * N;
        bytes += 4.0 * ((double)d.M * K + 3.0 * Cx * N + 2.0 * d.M * N);
    }
    a.x_floats = (int)((xmax + 3) & ~3LL);
    a.fault_inject = g_trunk_fault_inject;
    const bool wide = N > 16;
    const size_t lds = ((size_t)a.x_floats + 16 * 8 * 2 * 2 + kNetWaves * (wide ? 2 : 1) * 4 * kRedPitch + 16) * sizeof(float);
    if (lds > 160 * 1024) return MCVC_ERR_INVALID;
    mcvc_launch(zero_words_kernel, dim3(1), dim3(64), 0, s, ZeroWordsKArgs{a.sync, MCVC_TRUNK_SYNC_WORDS - 1});
    TraceScope ts(K_TRUNK, s, flops, bytes);
    static bool done[2] = {false, false};
    if (!done[wide]) {
        const void* fn = wide ? reinterpret_cast<const void*>(trunk_bwd_net_kernel<2>) : reinterpret_cast<const void*>(trunk_bwd_net_kernel<1>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done[wide] = true;
    }
    if (wide) mcvc_launch(trunk_bwd_net_kernel<2>, dim3(kNetGrid), dim3(kNetThreads), lds, s, a);
    else mcvc_launch(trunk_bwd_net_kernel<1>, dim3(kNetGrid), dim3(kNetThreads), lds, s, a);
    return (int)hipGetLastError();
}
