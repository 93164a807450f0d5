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
    int taps;              // KH * KW of the tensor this job reads (weight jobs; the fused update's tile geometry)
    long long xi_stride;   // PACK_WINO_*: floats between the 36 transformed matrices
};
struct PackPtrs { const float* p[128]; };
constexpr size_t kPackNetLds = 32 * 75 * sizeof(float);      // largest tap count (5x15) x 32 input channels; >= one 32x33 tile
int mcvc_pack_net_launch(const PackJob* d_jobs, int njobs, int nblocks, const PackDgradArgs* d_dga, const PackPtrs& ptrs, float* packed,
                         double bytes, hipStream_t s);
int mcvc_pack_trunk_t_launch(const float* w, float* dst, int Cout, int Cin, int KW, int ld, int co_off, hipStream_t s);

// ---- optimizer step fused with the re-pack (update_net_kernel): ONE workgroup owns a tile of filters of one parameter tensor -- it applies
// Adam to the tile (p, m, v in place, gradient cleared behind the read), keeps the updated weights in LDS and writes EVERY packed form the
// planner derives from that tensor (the "emits": the PackJob entries of this parameter) out of LDS.  One reader and one writer per weight:
// no second pass over the OIHW tensors, no ordering problem between the in-place update and the re-pack (train.py:242,299 is optimizer.step()).
#include "misc.h"
struct UpdOwner {
    int param;             // parameter-table index
    int block0, gx;        // first workgroup of this owner in the flat grid; tiles along ci
    int Cout, Cin, taps;   // tensor [Cout][Cin][taps]; flat owners (biases, norm affine parameters): Cout = numel, Cin = taps = 1
    int CB, IB;            // tile: CB output channels x IB input channels x all taps
    int e0, ne;            // emits: jobs [e0, e0 + ne) of the job table
    int flat;              // 1: plain elementwise owner, 256 elements per workgroup
    int vec4;              // 1: every tile row is a 16-byte-aligned run of a multiple of four floats
};
struct UpdAdam { long long d_g, d_g2, d_m, d_v; int has_g2, zero; AdamCoef c; };     // d_*: float offsets from a parameter to its gradient(s) / moments
// LDS row pitch of an owner tile: the run of IB * taps floats rounded up to four, + 1 -- ODD, so that the emits that walk the output channels
// (lane = co: the K-major forward copies, the forward Winograd sets) read 32 different banks; r4's pitch was a multiple of four (the Adam
// phase stored float4s) and those reads were 4-way conflicts -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.71 (r4 PMC pass).
static inline int mcvc_upd_pitch(int IB, int taps) { return ((IB * taps + 3) & ~3) + 1; }
constexpr size_t kUpdLds = 52 * 1024;
int mcvc_update_net_launch(const UpdOwner* d_owners, int nown, int nblocks, const PackJob* d_jobs, const PackDgradArgs* d_dga, const PackPtrs& ptrs,
                           float* packed, const UpdAdam& ad, double bytes, hipStream_t s);
