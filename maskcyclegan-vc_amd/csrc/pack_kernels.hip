// Weight re-packing for the direct-conv kernels (conv_kernels.hip).
//
// The nn.Module boundary keeps the reference's OIHW / OIW parameter tensors (state_dict drop-in,
// SURVEY.md Appendix B).  The MFMA kernels want K-major operands so that the 32 lanes of an A
// fragment read 32 consecutive output channels:
//   forward : Wf[(ci*KH+kh)*KW+kw][co]                       = W[co][ci][kh][kw]
//   dgrad   : Wd[cls][((co*NTH+u)*NTW+v)][ci]                = W[co][ci][kh(u)][kw(v)]
// where for stride 1 the single class is the 180-degree flipped kernel and for stride 2 the four
// output-parity classes (ih%2, iw%2) each keep the taps with matching parity -- so the
// data-gradient of a strided conv runs as four dense stride-1 convs with no zero-insertion.
// Packed buffers are zero-initialised once by the owner; the kernels only write valid entries, so
// channel/row padding stays zero.
#include "mcvc_common.h"
#include "pack.h"
#include "trace.h"
#include "launch.h"
#include "wino.h"
#include "wino4.h"

// [R=Cout][K] -> dst[k*ld + co_off + co], 32x32 LDS tiles, both sides coalesced
// taps > 0: the rows are re-ordered tap-major (k = ci * taps + tap  ->  row tap * Cin + ci)
__device__ __forceinline__ void pack_fwd_tile(const float* __restrict__ w, float* __restrict__ dst, int Cout, int K, int ld, int co_off,
                                              int bx, int by, float* smem, int taps = 0)
{
    float (*tile)[33] = reinterpret_cast<float (*)[33]>(smem);
    const int k0 = bx * 32, c0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = c0 + r, k = k0 + tx;
        tile[r][tx] = (co < Cout && k < K) ? w[(long long)co * K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, co = c0 + tx;
        if (k < K && co < Cout) {
            const int row = taps > 0 ? (k % taps) * (K / taps) + k / taps : k;
            dst[(long long)row * ld + co_off + co] = tile[tx][r];
        }
    }
}

// one block per (ci tile of 32, co): load W[co][ci0..ci0+32)[taps] (contiguous) and scatter rows of 32 ci
__device__ __forceinline__ void pack_dgrad_tile(const float* __restrict__ w, float* __restrict__ dst, const PackDgradArgs& a, int bx, int by,
                                                float* lds)
{
    const int ci0 = bx * 32, co = by;
    const int khkw = a.KH * a.KW;
    int nci = a.Cin - ci0; if (nci > 32) nci = 32;
    const float* src = w + ((long long)co * a.Cin + ci0) * khkw;
    for (int i = threadIdx.x; i < nci * khkw; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (int c = 0; c < a.ncls; ++c) {
        const int nt = a.cls[c].nth * a.cls[c].ntw;
        for (int t = grp; t < nt; t += 8) {
            const int u = t / a.cls[c].ntw, v = t - u * a.cls[c].ntw;
            const int kh = a.cls[c].khmax - a.step * u, kw = a.cls[c].kwmax - a.step * v;
            if (lane < nci) {
                const float wv = lds[lane * khkw + kh * a.KW + kw];
                if (a.tapmajor) {
                    dst[a.cls[c].offset + ((long long)t * a.cout_rows + a.co_off + co) * a.ld + ci0 + lane] = wv;
                } else if (a.merged) {
                    const long long row = ((long long)(a.co_off + co) * a.mg_kh + (u + a.cls[c].su)) * a.mg_kw + (v + a.cls[c].sv);
                    dst[row * a.ld + 4 * (ci0 + lane) + 2 * a.cls[c].qh + a.cls[c].qw] = wv;
                } else {
                    dst[a.cls[c].offset + ((long long)(a.co_off + co) * nt + t) * a.ld + ci0 + lane] = wv;
                }
            }
        }
    }
}

// Wt[ci][(co_off + co) * KW + kwp] = W[co][ci][KW-1-kwp]   (transposed + flipped copy for the fused trunk data-gradient)
__device__ __forceinline__ void pack_trunk_t_tile(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KW, int ld,
                                                  int co_off, int bx, int by, float* smem)
{
    float (*tile)[33] = reinterpret_cast<float (*)[33]>(smem);
    const int ci0 = bx * 32, co0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int kw = 0; kw < KW; ++kw) {
        for (int r = ty; r < 32; r += 8) {          // r = co, tx = ci
            const int co = co0 + r, ci = ci0 + tx;
            tile[r][tx] = (co < Cout && ci < Cin) ? w[((long long)co * Cin + ci) * KW + kw] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {          // r = ci, tx = co
            const int ci = ci0 + r, co = co0 + tx;
            if (ci < Cin && co < Cout) dst[(long long)ci * ld + (long long)(co_off + co) * KW + (KW - 1 - kw)] = tile[tx][r];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int K, int ld, int co_off)
{
    __shared__ float smem[32 * 33];
    pack_fwd_tile(w, dst, Cout, K, ld, co_off, blockIdx.x, blockIdx.y, smem);
}

__global__ void __launch_bounds__(256) pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ dst, PackDgradArgs a)
{
    extern __shared__ float lds[];
    pack_dgrad_tile(w, dst, a, blockIdx.x, blockIdx.y, lds);
}

__global__ void __launch_bounds__(256) pack_trunk_t_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KW, int ld,
                                                           int co_off)
{
    __shared__ float smem[32 * 33];
    pack_trunk_t_tile(w, dst, Cout, Cin, KW, ld, co_off, blockIdx.x, blockIdx.y, smem);
}

struct CopyKArgs { const float* src; float* dst; int n; };
__global__ void copy_kernel(const Twin<CopyKArgs> tw)
{
    const CopyKArgs ka_ = tw.v[blockIdx.z];
    const float* __restrict__ src = ka_.src;
    float* __restrict__ dst = ka_.dst;
    int n = ka_.n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// Whole-network re-pack in ONE launch: a device-resident job table maps each workgroup to (job, tile).  Replaces ~130
// tiny launches per generator (and their launch gaps) after every optimizer step.
struct PackNetKArgs { const PackJob* jobs; int njobs; const PackDgradArgs* dga; PackPtrs ptrs; float* packed; };
__global__ void __launch_bounds__(256) pack_net_kernel(const Twin<PackNetKArgs> tw)
{
    const PackNetKArgs& ka_ = tw.v[blockIdx.z];
    const PackJob* __restrict__ jobs = ka_.jobs;
    int njobs = ka_.njobs;
    const PackDgradArgs* __restrict__ dga = ka_.dga;
    const PackPtrs& ptrs = ka_.ptrs;
    float* __restrict__ packed = ka_.packed;
    extern __shared__ float lds[];
    const int blk = blockIdx.x;
    int lo = 0, hi = njobs - 1;                       // last job with block0 <= blk
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= blk) lo = mid; else hi = mid - 1; }
    const PackJob j = jobs[lo];
    const int rel = blk - j.block0;
    const int bx = rel % j.gx, by = rel / j.gx;
    const float* w = ptrs.p[j.param];
    float* dst = packed + j.dst;
    switch (j.kind) {
    case PACK_FWD: pack_fwd_tile(w, dst, j.Cout, j.K, j.ld, j.co_off, bx, by, lds); break;
    case PACK_FWD_TAP: pack_fwd_tile(w, dst, j.Cout, j.K, j.ld, j.co_off, bx, by, lds, j.KW); break;          // (KW field = taps per channel)
    case PACK_DGRAD: pack_dgrad_tile(w, dst, dga[j.dg], bx, by, lds); break;
    case PACK_TRUNK_T: pack_trunk_t_tile(w, dst, j.Cout, j.Cin, j.KW, j.ld, j.co_off, bx, by, lds); break;
    case PACK_WINO_F: wino_weight_tile(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, 0, bx, by); break;
    case PACK_WINO_D: wino_weight_tile(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, 1, bx, by); break;
    case PACK_WINO3_D: wino3_weight_tile(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, bx, by); break;
    case PACK_WINO3_F: wino3_weight_fwd_tile(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, bx, by); break;
    case PACK_WINO4_F: wino4_weight_tile(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, 0, bx, by); break;
    case PACK_WINO4_D: wino4_weight_tile(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, 1, bx, by); break;
    case PACK_WINO43_D: wino3_weight_tile_p<6>(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, bx, by); break;
    case PACK_WINO43_F: wino3_weight_fwd_tile_p<6>(w, dst, j.Cout, j.Cin, j.ld, j.xi_stride, j.co_off, bx, by); break;
    default: { const int i = rel * 256 + threadIdx.x; if (i < j.Cout) dst[i] = w[i]; } break;     // PACK_COPY
    }
}

// ---- optimizer step fused with the re-pack (pack.h: UpdOwner) -----------------------------------------------------------------
// LDS tile of updated weights: T[c][i * taps + t] at c * pitch, c < nc output channels, i < ni input channels of the tile at (co0, ci0)
struct UpdTile { const float* lds; int pitch, taps, co0, ci0, nc, ni, CB, IB, Cin; };

// K-major forward copy (rows k = ci * taps + t, or tap-major t * Cin + ci): runs of CB consecutive output channels
__device__ __forceinline__ void upd_emit_fwd(const PackJob& j, float* __restrict__ dst, const UpdTile& t, bool tapmajor)
{
    const int nk = t.ni * t.taps;
    for (int idx = threadIdx.x; idx < nk * t.CB; idx += 256) {
        const int c = idx % t.CB, kl = idx / t.CB;
        if (c >= t.nc) continue;
        const int i = kl / t.taps, tp = kl - i * t.taps;
        const int ci = t.ci0 + i;
        const long long row = tapmajor ? (long long)tp * t.Cin + ci : (long long)ci * t.taps + tp;
        dst[row * j.ld + j.co_off + t.co0 + c] = t.lds[c * t.pitch + kl];
    }
}

// data-gradient copies (pack_dgrad_tile's three layouts): runs of IB consecutive input channels
__device__ __forceinline__ void upd_emit_dgrad(float* __restrict__ dst, const PackDgradArgs& a, const UpdTile& t)
{
    for (int cl = 0; cl < a.ncls; ++cl) {
        const DgradClass& k = a.cls[cl];
        const int nt = k.nth * k.ntw;
        const int total = nt * t.nc * t.IB;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int i = idx % t.IB, r = idx / t.IB;
            if (i >= t.ni) continue;
            const int c = r % t.nc, tp = r / t.nc;
            const int u = tp / k.ntw, v = tp - u * k.ntw;
            const int kh = k.khmax - a.step * u, kw = k.kwmax - a.step * v;
            const float wv = t.lds[c * t.pitch + i * t.taps + kh * a.KW + kw];
            const int co = t.co0 + c, ci = t.ci0 + i;
            if (a.tapmajor) {
                dst[k.offset + ((long long)tp * a.cout_rows + a.co_off + co) * a.ld + ci] = wv;
            } else if (a.merged) {
                const long long row = ((long long)(a.co_off + co) * a.mg_kh + (u + k.su)) * a.mg_kw + (v + k.sv);
                dst[row * a.ld + 4 * ci + 2 * k.qh + k.qw] = wv;
            } else {
                dst[k.offset + ((long long)(a.co_off + co) * nt + tp) * a.ld + ci] = wv;
            }
        }
    }
}

// Wt[ci][(co_off + co) * KW + kwp] = W[co][ci][KW-1-kwp]: runs of CB * KW consecutive floats per input channel
__device__ __forceinline__ void upd_emit_trunk_t(const PackJob& j, float* __restrict__ dst, const UpdTile& t)
{
    const int KW = t.taps, span = t.nc * KW;
    for (int idx = threadIdx.x; idx < span * t.ni; idx += 256) {
        const int q = idx % span, i = idx / span;
        const int c = q / KW, kwp = q - c * KW;
        dst[(long long)(t.ci0 + i) * j.ld + (long long)(j.co_off + t.co0) * KW + q] = t.lds[c * t.pitch + i * KW + (KW - 1 - kwp)];
    }
}

// Winograd weight sets: one filter per thread; threads along co for the forward sets (columns = output channels), along ci for the
// data-gradient sets (columns = input channels)
template <class F>
__device__ __forceinline__ void upd_emit_filters(const UpdTile& t, bool along_co, F&& f)
{
    const int nf = t.nc * t.ni;
    for (int q = threadIdx.x; q < nf; q += 256) {
        const int c = along_co ? q % t.nc : q / t.ni, i = along_co ? q / t.nc : q % t.ni;
        f(t.lds + c * t.pitch + i * 25, t.co0 + c, t.ci0 + i);
    }
}

struct UpdNetKArgs { const UpdOwner* owners; int nown; const PackJob* jobs; const PackDgradArgs* dga; PackPtrs ptrs; float* packed; UpdAdam ad; };
__global__ void __launch_bounds__(256) update_net_kernel(const Twin<UpdNetKArgs> tw)
{
    const UpdNetKArgs& ka_ = tw.v[blockIdx.z];
    const UpdOwner* __restrict__ owners = ka_.owners;
    const PackJob* __restrict__ jobs = ka_.jobs;
    const PackDgradArgs* __restrict__ dga = ka_.dga;
    float* __restrict__ packed = ka_.packed;
    const UpdAdam ad = ka_.ad;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int blk = blockIdx.x;
    int lo = 0, hi = ka_.nown - 1;                    // last owner with block0 <= blk
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (owners[mid].block0 <= blk) lo = mid; else hi = mid - 1; }
    const UpdOwner o = owners[lo];
    const int rel = blk - o.block0;
    float* __restrict__ p = const_cast<float*>(ka_.ptrs.p[o.param]);
    float* __restrict__ g = p + ad.d_g;
    float* __restrict__ g2 = ad.has_g2 ? p + ad.d_g2 : nullptr;
    float* __restrict__ m = p + ad.d_m;
    float* __restrict__ v = p + ad.d_v;
    const AdamCoef c = ad.c;
    if (o.flat) {                                     // biases, norm affine parameters: 256 elements per workgroup, at most a copy to emit
        const int i = rel * 256 + threadIdx.x;
        if (i >= o.Cout) return;
        float gr = g[i];
        if (g2) gr += g2[i];
        if (ad.zero) { g[i] = 0.f; if (g2) g2[i] = 0.f; }
        float pn = p[i], mn = m[i], vn = v[i];
        adam_elem(pn, gr, mn, vn, c);
        p[i] = pn; m[i] = mn; v[i] = vn;
        for (int e = 0; e < o.ne; ++e) packed[jobs[o.e0 + e].dst + i] = pn;      // (PACK_COPY)
        return;
    }
    const int bx = rel % o.gx, by = rel / o.gx;
    UpdTile t;
    t.lds = lds; t.taps = o.taps; t.CB = o.CB; t.IB = o.IB; t.Cin = o.Cin;
    t.co0 = by * o.CB; t.ci0 = bx * o.IB;
    t.nc = min(o.CB, o.Cout - t.co0); t.ni = min(o.IB, o.Cin - t.ci0);
    const int run = t.ni * o.taps;                    // floats per tile row (contiguous in the OIHW tensor)
    t.pitch = ((o.IB * o.taps + 3) & ~3) + 1;                 // (mcvc_upd_pitch: odd)
    const long long row_stride = (long long)o.Cin * o.taps;
    const long long base = (long long)t.co0 * row_stride + (long long)t.ci0 * o.taps;
    // ---- Adam on the tile; the new weights stay in LDS
    if (o.vec4) {
        const int run4 = run >> 2, total4 = t.nc * run4;
        // two float4 per thread and round: all ten loads of a round are in flight before the first is used (the tile is one DRAM latency
        // deep otherwise: a thread of a 32 x 16 x 25 tile walks 12 dependent load -> compute -> store rounds)
        constexpr int U = 2;
        for (int q0 = threadIdx.x; q0 < total4; q0 += U * 256) {
            float4 pp[U], gg[U], hh[U], mm[U], vv[U];
            long long off[U]; int ldo[U]; bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * 256;
                ok[u] = q < total4;
                const int qq = ok[u] ? q : 0;
                const int cc = qq / run4, r4 = qq - cc * run4;
                off[u] = base + cc * row_stride + 4 * r4;
                ldo[u] = cc * t.pitch + 4 * r4;
                pp[u] = *reinterpret_cast<const float4*>(p + off[u]);
                gg[u] = *reinterpret_cast<const float4*>(g + off[u]);
                hh[u] = g2 ? *reinterpret_cast<const float4*>(g2 + off[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
                mm[u] = *reinterpret_cast<const float4*>(m + off[u]);
                vv[u] = *reinterpret_cast<const float4*>(v + off[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                if (g2) { gg[u].x += hh[u].x; gg[u].y += hh[u].y; gg[u].z += hh[u].z; gg[u].w += hh[u].w; }
                if (ad.zero) {
                    *reinterpret_cast<float4*>(g + off[u]) = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g2) *reinterpret_cast<float4*>(g2 + off[u]) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float* pf = reinterpret_cast<float*>(&pp[u]);
                float* gf = reinterpret_cast<float*>(&gg[u]);
                float* mf = reinterpret_cast<float*>(&mm[u]);
                float* vf = reinterpret_cast<float*>(&vv[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k) adam_elem(pf[k], gf[k], mf[k], vf[k], c);
                *reinterpret_cast<float4*>(p + off[u]) = pp[u];
                *reinterpret_cast<float4*>(m + off[u]) = mm[u];
                *reinterpret_cast<float4*>(v + off[u]) = vv[u];
                lds[ldo[u]] = pp[u].x; lds[ldo[u] + 1] = pp[u].y; lds[ldo[u] + 2] = pp[u].z; lds[ldo[u] + 3] = pp[u].w;      // (odd pitch: no 16-byte alignment)
            }
        }
    } else {
        const int total = t.nc * run;
        for (int q = threadIdx.x; q < total; q += 256) {
            const int cc = q / run, r = q - cc * run;
            const long long off = base + cc * row_stride + r;
            float gr = g[off];
            if (g2) gr += g2[off];
            if (ad.zero) { g[off] = 0.f; if (g2) g2[off] = 0.f; }
            float pn = p[off], mn = m[off], vn = v[off];
            adam_elem(pn, gr, mn, vn, c);
            p[off] = pn; m[off] = mn; v[off] = vn;
            lds[cc * t.pitch + r] = pn;
        }
    }
    if (o.ne == 0) return;
    __syncthreads();
    // ---- every packed form of this tile, out of LDS
    for (int e = 0; e < o.ne; ++e) {
        const PackJob j = jobs[o.e0 + e];
        float* dst = packed + j.dst;
        switch (j.kind) {
        case PACK_FWD: upd_emit_fwd(j, dst, t, false); break;
        case PACK_FWD_TAP: upd_emit_fwd(j, dst, t, true); break;
        case PACK_DGRAD: upd_emit_dgrad(dst, dga[j.dg], t); break;
        case PACK_TRUNK_T: upd_emit_trunk_t(j, dst, t); break;
        case PACK_WINO_F: upd_emit_filters(t, true, [&](const float* f, int co, int ci) { wino_weight_core(f, dst, co, ci, j.ld, j.xi_stride, j.co_off, 0); }); break;
        case PACK_WINO_D: upd_emit_filters(t, false, [&](const float* f, int co, int ci) { wino_weight_core(f, dst, co, ci, j.ld, j.xi_stride, j.co_off, 1); }); break;
        case PACK_WINO3_D: upd_emit_filters(t, false, [&](const float* f, int co, int ci) { wino3_weight_core_p<4>(f, dst, co, ci, j.ld, j.xi_stride, j.co_off); }); break;
        case PACK_WINO3_F: upd_emit_filters(t, true, [&](const float* f, int co, int ci) { wino3_weight_fwd_core_p<4>(f, dst, co, ci, j.ld, j.xi_stride, j.co_off); }); break;
        case PACK_WINO4_F: upd_emit_filters(t, true, [&](const float* f, int co, int ci) { wino4_weight_core(f, dst, co, ci, j.ld, j.xi_stride, j.co_off, 0); }); break;
        case PACK_WINO4_D: upd_emit_filters(t, false, [&](const float* f, int co, int ci) { wino4_weight_core(f, dst, co, ci, j.ld, j.xi_stride, j.co_off, 1); }); break;
        case PACK_WINO43_D: upd_emit_filters(t, false, [&](const float* f, int co, int ci) { wino3_weight_core_p<6>(f, dst, co, ci, j.ld, j.xi_stride, j.co_off); }); break;
        case PACK_WINO43_F: upd_emit_filters(t, true, [&](const float* f, int co, int ci) { wino3_weight_fwd_core_p<6>(f, dst, co, ci, j.ld, j.xi_stride, j.co_off); }); break;
        default: break;
        }
    }
}

int mcvc_update_net_launch(const UpdOwner* d_owners, int nown, int nblocks, const PackJob* d_jobs, const PackDgradArgs* d_dga, const PackPtrs& ptrs,
                           float* packed, const UpdAdam& ad, double bytes, hipStream_t s)
{
    TraceScope ts(K_ADAM, s, 0.0, bytes);
    mcvc_launch(update_net_kernel, dim3((unsigned)nblocks), dim3(256), kUpdLds, s, UpdNetKArgs{d_owners, nown, d_jobs, d_dga, ptrs, packed, ad});
    return (int)hipGetLastError();
}

int mcvc_pack_net_launch(const PackJob* d_jobs, int njobs, int nblocks, const PackDgradArgs* d_dga, const PackPtrs& ptrs, float* packed,
                         double bytes, hipStream_t s)
{
    TraceScope ts(K_PACK, s, 0.0, bytes);
    mcvc_launch(pack_net_kernel, dim3((unsigned)nblocks), dim3(256), kPackNetLds, s, PackNetKArgs{d_jobs, njobs, d_dga, ptrs, packed});
    return (int)hipGetLastError();
}

int mcvc_pack_trunk_t_launch(const float* w, float* dst, int Cout, int Cin, int KW, int ld, int co_off, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(Cin, 32), (unsigned)cdiv_i(Cout, 32));
    TraceScope ts(K_PACK, s, 0.0, 8.0 * (double)Cout * Cin * KW);
    hipLaunchKernelGGL(pack_trunk_t_kernel, grid, dim3(256), 0, s, w, dst, Cout, Cin, KW, ld, co_off);
    return (int)hipGetLastError();
}

int mcvc_pack_fwd_launch(const float* w, float* dst, int Cout, int K, int ld, int co_off, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(K, 32), (unsigned)cdiv_i(Cout, 32));
    TraceScope ts(K_PACK, s, 0.0, 8.0 * (double)Cout * K);
    hipLaunchKernelGGL(pack_fwd_kernel, grid, dim3(256), 0, s, w, dst, Cout, K, ld, co_off);
    return (int)hipGetLastError();
}

int mcvc_pack_dgrad_launch(const float* w, float* dst, const PackDgradArgs& a, int Cout, hipStream_t s)
{
    dim3 grid((unsigned)cdiv_i(a.Cin, 32), (unsigned)Cout);
    const size_t lds = (size_t)32 * a.KH * a.KW * sizeof(float);
    TraceScope ts(K_PACK, s, 0.0, 8.0 * (double)Cout * a.Cin * a.KH * a.KW);
    hipLaunchKernelGGL(pack_dgrad_kernel, grid, dim3(256), lds, s, w, dst, a);
    return (int)hipGetLastError();
}

int mcvc_copy_launch(const float* src, float* dst, int n, hipStream_t s)
{
    TraceScope ts(K_ELEMENTWISE, s, 0.0, 8.0 * n);
    mcvc_launch(copy_kernel, dim3((unsigned)cdiv_i(n, 256)), dim3(256), 0, s, CopyKArgs{src, dst, n});
    return (int)hipGetLastError();
}
