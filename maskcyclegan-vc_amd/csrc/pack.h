// Data-gradient class descriptors shared by the packer and the network planner.
#pragma once
#include <hip/hip_runtime.h>

struct DgradClass {
    int nth, ntw;          // taps of this output-parity class
    int khmax, kwmax;      // tap u -> kh = khmax - step*u
    int pad_h, pad_w;      // padding of the equivalent stride-1 conv over dY
    int qh, qw;            // output parity (0 for stride 1)
    int su, sv;            // merged (stride 2) layout: tap (u, v) of this class sits at (u + su, v + sv) of the common window
    long long offset;      // float offset of this class inside the layer's dgrad pack
};

struct PackDgradArgs {
    int Cin, KH, KW;
    int step;              // = conv stride (1 or 2)
    int ld;                // row length (Cin rounded up to 32)
    int co_off;            // channel offset of this branch inside the concatenated (value|gate) conv
    int merged;            // 1: one matrix for all parity classes: row (co, u', v'), column 4*ci + 2*qh + qw
    int mg_kh, mg_kw;      // common tap window of the merged layout
    int ncls;
    DgradClass cls[4];
    int tapmajor;          // 1: rows (t, co) instead of (co, t): row = t * cout_rows + co_off + co (the implicit GEMM's per-class A operand)
    int cout_rows;         // rows per tap (= concatenated output channels)
};

int mcvc_pack_fwd_launch(const float* w, float* dst, int Cout, int K, int ld, int co_off, hipStream_t s);
int mcvc_pack_dgrad_launch(const float* w, float* dst, const PackDgradArgs& a, int Cout, hipStream_t s);
int mcvc_copy_launch(const float* src, float* dst, int n, hipStream_t s);

// ---- whole-network re-pack (one launch): device-resident job table
enum PackKind { PACK_FWD = 0, PACK_DGRAD = 1, PACK_COPY = 2, PACK_TRUNK_T = 3, PACK_WINO_F = 4, PACK_WINO_D = 5, PACK_WINO3_D = 6, PACK_WINO3_F = 7, PACK_WINO4_F = 8, PACK_WINO4_D = 9, PACK_WINO43_D = 10, PACK_WINO43_F = 11,
                PACK_FWD_TAP = 12 };      // forward K-major copy with TAP-major rows: dst[(tap * Cin + ci) * ld + co]   (implicit GEMM, sgemm.h)
struct PackJob {
    int kind, param;       // param = index into the parameter-pointer table
    int block0, gx;        // first workgroup of this job in the flat grid; tile (bx, by) = (rel % gx, rel / gx)
    long long dst;         // float offset inside the packed buffer
    int Cout, K, ld, co_off, KW, Cin;
    int dg;                // PACK_DGRAD: index into the PackDgradArgs table
    int pad_;
    long long xi_stride;   // PACK_WINO_*: floats between the 36 transformed matrices
};
struct PackPtrs { const float* p[128]; };
constexpr size_t kPackNetLds = 32 * 75 * sizeof(float);      // largest tap count (5x15) x 32 input channels; >= one 32x33 tile
int mcvc_pack_net_launch(const PackJob* d_jobs, int njobs, int nblocks, const PackDgradArgs* d_dga, const PackPtrs& ptrs, float* packed,
                         double bytes, hipStream_t s);
int mcvc_pack_trunk_t_launch(const float* w, float* dst, int Cout, int Cin, int KW, int ld, int co_off, hipStream_t s);
