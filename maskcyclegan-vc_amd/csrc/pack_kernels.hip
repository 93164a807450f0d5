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
