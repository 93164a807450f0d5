// Network-level schedule of the MaskCycleGAN-VC Generator / Discriminator on the gfx950 kernels and
// the exported C ABI (include/mcvc.h).  The library owns the layer schedule so that one host call
// launches a whole forward or backward pass back-to-back on the caller's HIP stream (graph-capturable:
// no allocation, no synchronisation, no host-side state).
//
// Layer walk follows mask_cyclegan_vc/model.py:239-280 (Generator.forward) and :340-349
// (Discriminator.forward); parameter indices follow named_parameters() order.
//
// Internal layouts (all fp32):
//   2-D activations  NCHW  [B][C][H][W]
//   1-D trunk        [C][B][T4]  -- channel-major with the batch inside, so every Conv1d is a
//                    single-image KHx1 convolution over an image of B rows; the reference's
//                    view(B, 5120, 1, -1) (model.py:249-251, channel = c*20 + h) and view(B,256,20,-1)
//                    (:270-271) become pure stride changes folded into the producing kernels.
//   value|gate pairs are one convolution with concatenated output channels [value C | gate C].
#include "mcvc_common.h"
#include "twin.h"
#include "pack.h"
#include "misc.h"
#include "trunk.h"
#include "wino.h"
#include "wino4.h"
#include "sgemm.h"
#include "sampler.h"
#include "../../include/mcvc.h"
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

static int g_deterministic = [] { const char* e = getenv("MCVC_DETERMINISTIC"); return (e && atoi(e) != 0) ? 1 : 0; }();
int mcvc_deterministic() { return g_deterministic; }

namespace {

constexpr float kInEps = 1e-5f;

struct View { float* p; long long sb, sc; int sh; };
struct CView { const float* p; long long sb, sc; int sh; };
static inline CView cv(const View& v) { return CView{v.p, v.sb, v.sc, v.sh}; }

struct Exec {
    hipStream_t s;
    bool dry;
    int err;
    bool fine_ms = false;          // generator backward: four milestones (flags & MCVC_BWD_FINE_MILESTONES), see gen_backward_impl
    float* slabs;
    long long slab_cap;
    long long slab_need;
    int max_split;                 // 0 = planner default (<= 64)
    // weight gradients are off the critical path of a backward pass: with an auxiliary stream they run beside the
    // data-gradient chain.  `readers` remembers, per dY buffer, the event after which its last aux-stream reader is done.
    hipStream_t s2;
    float* wslabs;
    long long wslab_cap;
    long long wslab_need;
    float* wv; float* wm;          // Winograd scratch: transformed input V[36][K][tiles], products M[36][M][tiles]
    long long wino_cap;            // floats available in each
    float* wv2; float* wm2; float* wu;   // the weight-gradient's own set (it runs on the auxiliary stream beside the dgrad): Vt, dMt, dU
    long long wu_cap;
    float* sg; long long sg_cap, sg_need;    // staging region of the staged-GEMM convolutions (sgemm.h): forward / data gradient (main stream)
    float* sgw; long long sgw_cap, sgw_need; // ... and the weight gradients' own (they may run on the auxiliary stream beside a data gradient)
    const float* const* params;    // the pass's parameter table (that path's data gradient multiplies the OIHW tensors themselves)
    int br1_on_main = 0;           // conv_wgrad: run the gate branch's generic weight gradient on the main stream (see there)
    int no_join = 0;               // backward pass: leave the auxiliary stream un-joined at the end (mcvc_gen_backward_flags)
    int fuse_next = 0;             // the caller's next step is a norm that can absorb a Winograd output transform (set before conv_fwd)
    int pend_pts = 0;              // 16 / 36 / 43 (= F(4x4,3x3)) / 64: the output transform in `pend` has not run yet -- norm_fwd runs it (fused when it fits)
    WinoOutArgs pend;
    int y_xs_pw = 0; long long y_xs_plane = 0;   // the next norm_fwd writes y in the phase-split padded layout (sgemm.h); consumed by norm_fwd
    int dx_pitch = 0;              // the next norm_bwd writes dx in the padded dY layout (rows of dx_pitch floats + a zero row); consumed by norm_bwd
    int wgrad_x_xs = 0;            // conv_wgrad: x is in the phase-split padded layout
    int force_scheme = 0;          // op-level entries (mcvc_layer_*): 0 planner's choice, 1 Winograd with 2x2 output tiles only, 2 with 4x4 tiles
                                   // (thresholds on samples / tiles lifted), 3 no Winograd, 4 no Winograd and no staged GEMM (direct kernels)
    int pack_skips;                // what the last re-pack of `packed` left stale: bit 0 = generic trunk copies, bit 1 = direct copies of the Winograd layers
    unsigned* sync;                // arrival counters of the persistent trunk kernels (MCVC_TRUNK_SYNC_WORDS words of the scratch)
    std::vector<std::pair<const void*, hipEvent_t>> readers;
    void fail(int e) { if (!err && e) err = e; }
};

// What the last re-pack of a packed-weight buffer skipped (mcvc_gen_pack_small_batch refreshes only the copies the small-batch schedule
// reads).  The run-time dispatch predicates and the pack predicate are written separately; if they ever diverge, the generic path would
// read zero / stale weights silently -- so every pass looks its buffer up and the generic paths refuse to run on a skipped region.
static std::mutex g_pack_mu;
static std::map<const void*, int> g_pack_skips;
static void set_pack_skips(const void* packed, int skips) { std::lock_guard<std::mutex> lk(g_pack_mu); g_pack_skips[packed] = skips; }
static int get_pack_skips(const void* packed)
{
    std::lock_guard<std::mutex> lk(g_pack_mu);
    auto it = g_pack_skips.find(packed);
    return it == g_pack_skips.end() ? 0 : it->second;          // never packed through the small-batch entry: assume a full pack
}

// small pool of timing-less events, reused round-robin (a backward pass uses a few dozen; the pool is far larger, so an
// event is never re-recorded while a wait on its previous use can still be pending)
static hipEvent_t pool_event()
{
    static std::mutex mu;
    static std::vector<hipEvent_t> pool;
    static size_t next = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (pool.empty()) {            // the whole pool is created on first use (outside any stream capture: the warm-up iterations)
        pool.resize(1024);
        for (hipEvent_t& e : pool) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
    next = (next + 1) % pool.size();
    return pool[next];
}

// the main stream is about to overwrite `buf`: wait for aux-stream kernels that still read it
static void wait_readers(Exec& ex, const void* buf)
{
    if (!ex.s2 || ex.dry) return;
    for (auto it = ex.readers.begin(); it != ex.readers.end();) {
        if (it->first == buf) { ex.fail(mcvc_stream_wait(ex.s, it->second)); it = ex.readers.erase(it); }
        else ++it;
    }
}

static void join_aux(Exec& ex)
{
    if (!ex.s2 || ex.dry) return;
    hipEvent_t e = pool_event();
    ex.fail(mcvc_event_record(e, ex.s2));
    ex.fail(mcvc_stream_wait(ex.s, e));
    ex.readers.clear();
}

// -------------------------------------------------------------------------------------------------
struct ConvSpec {
    int Cin, Cout, nbr, KH, KW, stride, ph, pw;
    int wi[2], bi[2];
    int has_bias_grad;
    // derived
    int cout_tot, cout_pk, cin_pad, w_rows, cin_pk, dg_rows_co;
    long long off_fwd, off_bias, off_dgrad;
    // 5x5 stride-1 single-branch convs (upSample1/2): Winograd F(2x2,5x5) weight sets U[36][K+1][ld], forward and data-gradient
    int wino; long long off_wf, off_wd, wf_xi, wd_xi;
    long long off_w4f, off_w4d;    // ... and their F(4x4,5x5) twins U[64][...] (wino4.h; same row / column conventions and strides per point)
    // 5x5 stride-2 convs (downSample1/2): their merged 3x3 data-gradient as Winograd F(2x2,3x3): U[16][co (+1)][mg_ld]
    int wino3; long long off_w3, w3_xi;
    long long off_w3f, w3f_xi;     // forward twin over the four input phases: U[16][4*Cin (+1)][cout_pk]
    long long off_w43, off_w43f;   // ... and the 36-point F(4x4,3x3) sets of both (wino43_kernels.hip), same strides per point
    long long off_tk;          // KH == 1 convs: transposed + flipped [Cin][cout_tot*KW] copy for the fused small-batch trunk dgrad
    int ncls;
    DgradClass cls[4];
    // stride 2: the four output-parity classes are ONE stride-1 conv with 4*Cin output channels (4*ci + 2*qh + qw) whose
    // taps are zero-padded to a common window; its PixelShuffle(2) store scatters straight into dX[ci][2a+qh][2b+qw]
    int merged, mg_kh, mg_kw, mg_pad_h, mg_pad_w, mg_ld;
    // 3x3 stride-2 convs (the discriminators') at large batch: the merged matrix multiplies 16 tap slots of which 9 are non-zero; with
    // enough pixels to fill the chip per launch the four parity classes run as four exact stride-1 convs instead (1.78x fewer MACs)
    long long off_dcls;            // -1: no per-class copies
    DgradClass ucls[4];
    // 3x3 stride-2 padding-1 layers (the discriminators'): implicit-GEMM operands (sgemm.h) -- the forward copy with tap-major rows and the
    // four parity classes' data-gradient matrices with rows (tap, co)
    int igemm; long long off_ifwd, off_idg; DgradClass icls[4];
};

static void spec_finalize(ConvSpec& c, long long& cur)
{
    c.cout_tot = c.Cout * c.nbr;
    c.cout_pk = round_up_i(c.cout_tot, 32);
    c.cin_pad = round_up_i(c.Cin, 2);
    c.w_rows = c.cin_pad * c.KH * c.KW;
    c.off_fwd = cur; cur += (long long)(c.w_rows + 1) * c.cout_pk;      // + one all-zero pad row (DMA target for k >= K)
    c.off_bias = cur; cur += c.cout_pk;
    c.cin_pk = round_up_i(c.Cin, 32);
    c.dg_rows_co = round_up_i(c.cout_tot, 2);
    c.off_dgrad = cur;
    c.ncls = 0;
    const int st = c.stride;
    for (int qh = 0; qh < st; ++qh) {
        for (int qw = 0; qw < st; ++qw) {
            DgradClass k{};
            const int khmin = (qh + c.ph) % st, kwmin = (qw + c.pw) % st;
            if (khmin > c.KH - 1 || kwmin > c.KW - 1) continue;
            k.khmax = khmin + st * ((c.KH - 1 - khmin) / st);
            k.kwmax = kwmin + st * ((c.KW - 1 - kwmin) / st);
            k.nth = (k.khmax - khmin) / st + 1;
            k.ntw = (k.kwmax - kwmin) / st + 1;
            k.pad_h = (k.khmax - qh - c.ph) / st;
            k.pad_w = (k.kwmax - qw - c.pw) / st;
            k.qh = qh; k.qw = qw;
            k.offset = cur - c.off_dgrad;
            if (st == 1) cur += ((long long)c.dg_rows_co * k.nth * k.ntw + 1) * c.cin_pk;   // + zero pad row
            c.cls[c.ncls++] = k;
        }
    }
    c.merged = 0;
    if (st == 2) {
        c.merged = 1;
        c.mg_pad_h = c.mg_pad_w = 0;
        for (int k = 0; k < c.ncls; ++k) {
            if (c.cls[k].pad_h > c.mg_pad_h) c.mg_pad_h = c.cls[k].pad_h;
            if (c.cls[k].pad_w > c.mg_pad_w) c.mg_pad_w = c.cls[k].pad_w;
        }
        c.mg_kh = c.mg_kw = 1;
        for (int k = 0; k < c.ncls; ++k) {
            DgradClass& d = c.cls[k];
            d.su = c.mg_pad_h - d.pad_h; d.sv = c.mg_pad_w - d.pad_w; d.offset = 0;
            if (d.su + d.nth > c.mg_kh) c.mg_kh = d.su + d.nth;
            if (d.sv + d.ntw > c.mg_kw) c.mg_kw = d.sv + d.ntw;
        }
        c.mg_ld = round_up_i(4 * c.Cin, 32);
        cur += ((long long)c.dg_rows_co * c.mg_kh * c.mg_kw + 1) * c.mg_ld;                  // + zero pad row
    }
    c.off_dcls = -1;
    if (st == 2 && c.KH == 3 && c.KW == 3 && c.Cin >= 64) {
        cur = (cur + 3) & ~3LL;
        c.off_dcls = cur;
        for (int k = 0; k < c.ncls; ++k) {
            c.ucls[k] = c.cls[k];
            c.ucls[k].offset = cur - c.off_dcls;
            cur += ((long long)c.dg_rows_co * c.cls[k].nth * c.cls[k].ntw + 1) * c.cin_pk;   // + zero pad row
        }
    }
    c.igemm = 0; c.off_ifwd = c.off_idg = -1;
    if (st == 2 && c.KH == 3 && c.KW == 3 && c.ph == 1 && c.pw == 1 && (c.Cin % 64) == 0 && (c.cout_tot % 64) == 0 && c.ncls == 4) {
        c.igemm = 1;
        cur = (cur + 3) & ~3LL;
        c.off_ifwd = cur; cur += (long long)9 * c.Cin * c.cout_pk;
        // (the implicit data gradient reads this tap-major forward copy row-major -- sgemm.h arow -- its per-class copies of r4 are gone)
        for (int k = 0; k < c.ncls; ++k) { c.icls[k] = c.cls[k]; c.icls[k].offset = 0; }
    }
    cur = (cur + 3) & ~3LL;
    c.wino = (c.KH == 5 && c.KW == 5 && st == 1 && c.nbr == 1 && c.Cin >= 64 && c.Cout >= 64) ? 1 : 0;
    c.off_wf = c.off_wd = c.off_w4f = c.off_w4d = -1; c.wf_xi = c.wd_xi = 0;
    if (c.wino) {
        c.wf_xi = (long long)(c.cin_pad + 1) * c.cout_pk;                          // rows ci (+ zero pad row), columns co
        c.off_wf = cur; cur += 36 * c.wf_xi;
        c.wd_xi = (long long)(c.dg_rows_co + 1) * c.cin_pk;                       // rows co, columns ci
        c.off_wd = cur; cur += 36 * c.wd_xi;
        cur = (cur + 3) & ~3LL;
        c.off_w4f = cur; cur += 64 * c.wf_xi;
        c.off_w4d = cur; cur += 64 * c.wd_xi;
        cur = (cur + 3) & ~3LL;
    }
    c.wino3 = (c.merged && c.KH == 5 && c.KW == 5 && c.ph == 2 && c.pw == 2 && (4 * c.Cin) % 128 == 0 && c.cout_tot % 16 == 0) ? 1 : 0;
    c.off_w3 = -1; c.w3_xi = 0;
    c.off_w3f = -1; c.w3f_xi = 0;
    c.off_w43 = c.off_w43f = -1;
    if (c.wino3) {
        c.w3_xi = (long long)(c.dg_rows_co + 1) * c.mg_ld; c.off_w3 = cur; cur += 16 * c.w3_xi; cur = (cur + 3) & ~3LL;
        c.w3f_xi = (long long)(4 * c.Cin + 1) * c.cout_pk; c.off_w3f = cur; cur += 16 * c.w3f_xi; cur = (cur + 3) & ~3LL;
        c.off_w43 = cur; cur += 36 * c.w3_xi; cur = (cur + 3) & ~3LL;
        c.off_w43f = cur; cur += 36 * c.w3f_xi; cur = (cur + 3) & ~3LL;
    }
    c.off_tk = -1;
    if (c.KH == 1 && st == 1) { c.off_tk = cur; cur += (long long)c.Cin * c.cout_tot * c.KW; cur = (cur + 3) & ~3LL; }
}

static ConvSpec mk(int Cin, int Cout, int nbr, int KH, int KW, int stride, int ph, int pw, int w0, int b0, int w1, int b1, int bias_grad)
{
    ConvSpec c{};
    c.Cin = Cin; c.Cout = Cout; c.nbr = nbr; c.KH = KH; c.KW = KW; c.stride = stride; c.ph = ph; c.pw = pw;
    c.wi[0] = w0; c.bi[0] = b0; c.wi[1] = w1; c.bi[1] = b1; c.has_bias_grad = bias_grad;
    return c;
}

static inline int conv_out(int H, int K, int s, int p) { return (H + 2 * p - K) / s + 1; }

// ---- conv wrappers ---------------------------------------------------------------------------------
static void run_conv(Exec& ex, const ConvProblem& p, int NB, ConvIO io, long long y_total, const float* w, int w_rows, int w_cout,
                     const float* bias, int allow_split, int force_nsplit, int* nsplit_out)
{
    int ns = 1;
    if (force_nsplit > 0) ns = force_nsplit;
    else if (allow_split) ns = mcvc_conv_plan_nsplit(p, NB, 1);
    if (ns < 1) { ex.fail(MCVC_ERR_INVALID); ns = 1; }
    if (ex.max_split > 0 && ns > ex.max_split) ns = ex.max_split;
    if (io.accumulate && mcvc_deterministic()) ns = 1;          // a K-split accumulate adds with atomics
    if (!io.accumulate) {
        const long long need = (long long)(ns - 1) * y_total;
        if (need > ex.slab_need) ex.slab_need = need;
        if (!ex.dry && need > ex.slab_cap) { ex.fail(MCVC_ERR_WORKSPACE); return; }
    }
    if (nsplit_out) *nsplit_out = io.accumulate ? 1 : ns;
    if (ex.dry) return;
    io.slabs = ex.slabs; io.slab_stride = y_total; io.nsplit = ns;
    ex.fail(mcvc_conv_launch(p, NB, io, w, w_rows, w_cout, bias, ex.s, nullptr));
}

static bool wino_env_enabled()
{
    static const int en = mcvc_knob("MCVC_WINO", 1);
    return en != 0;
}
static thread_local int t_no_wino = 0;          // (op-level entries that ask for the direct / staged-GEMM kernels)
// Precise mode (mcvc_set_precise / env MCVC_PRECISE): every 5x5 layer on the direct kernels -- no Winograd scheme anywhere.  The fast default
// trades rounding for multiplies (after four Adam steps its parameters sit 2-3x further from an fp64 run than the
// reference's own fp32 arithmetic does; precise mode sits where the reference sits).  Process-wide; set it BEFORE networks are packed and
// engines are built (workspaces and packed copies are planned for the mode in force; a later switch fails loudly on a stale copy).
static int g_precise = [] { const char* e = getenv("MCVC_PRECISE"); return (e && atoi(e) != 0) ? 1 : 0; }();
static bool wino_enabled() { return wino_env_enabled() && !t_no_wino && !g_precise; }

// tile of the 36 batched products: 128 channels x 64 tiles measured best on all four shapes (128x128 wastes the ragged
// tile count of upSample1, 256x32 re-reads V too often); knob: MCVC_WINO_CFG = planner index + 1
static int wino_tile_cfg(int M, long long NT)
{
    static const int knob = mcvc_knob("MCVC_WINO_CFG", -1);
    (void)M; (void)NT;
    return knob >= 0 ? knob : 2;
}

// Samples per Winograd pass: the V / M workspaces hold at most kWinoMaxTiles tiles; a larger batch runs in equal chunks of samples
// (every op of the three-launch pipeline is per sample; weight gradients add up over the chunks).  0 = not applicable.
constexpr long long kWinoMaxTiles = 16384;
static int wino_chunk(int NB, long long tiles_per_sample)
{
    if (tiles_per_sample <= 0 || tiles_per_sample > kWinoMaxTiles) return 0;
    const long long nbmax = kWinoMaxTiles / tiles_per_sample;
    if (NB <= nbmax) return NB;
    const long long nch = (NB + nbmax - 1) / nbmax;
    return (int)((NB + nch - 1) / nch);
}
static long long wino_chunk_tiles(int NB, long long tiles_per_sample)          // padded tile count of the largest chunk
{
    const int nbc = wino_chunk(NB, tiles_per_sample);
    return nbc ? ((nbc * tiles_per_sample + 31) & ~31LL) : 0;
}

// 5x5 stride-1 conv as Winograd F(2x2,5x5): input transform -> 36 batched [M x K] x [K x tiles] products (one launch of the
// direct-conv kernel as a 1x1 conv over 36 images with per-image weights) -> output transform (+bias, PixelShuffle store).
// dgrad: the same on the flipped / transposed weight set; K = conv output channels, M = conv input channels.
// F(4x4,5x5) (wino4.h) from `MCVC_WINO4_NB` samples per pass on images whose sides are multiples of 4, when the packed buffer holds the
// 64-point weight sets (pack_skips bit 16 clear): 2.25x fewer multiplies and 2.25x smaller V / M than F(2x2,5x5).  Measured (ms/step, off vs on):
// bs=1 6.77 / 6.64 (upSample2 only: upSample1 has 20 tiles per sample), bs=2 10.8 / 10.3, bs=4 17.5 / 15.1, bs=8 31.1 / 24.3, bs=32 118.1 / 84.5
// (the last three with F(4x4,3x3) on downSample1/2 as well, MCVC_WINO43_NB: at one sample per pass that one costs 0.3 ms)
static int wino4_min_nb()
{
    static const int nb = mcvc_knob("MCVC_WINO4_NB", 1);
    return nb;
}
// fewest F(4x4,5x5) tiles in a pass (the GEMM's column count; 32-column granularity)
static int wino4_min_tiles()
{
    static const int n = mcvc_knob("MCVC_WINO4_MIN_TILES", 64);
    return n < 32 ? 32 : n;
}
static int wino43_min_nb()
{
    static const int nb = mcvc_knob("MCVC_WINO43_NB", 4);
    return nb;
}
static bool wino4_applies(const Exec& ex, const ConvSpec& c, int NB, int H, int W)
{
    if (ex.force_scheme == 1 || ex.force_scheme >= 3) return false;
    if (ex.force_scheme == 2) return c.wino && c.off_w4f >= 0 && (H & 3) == 0 && (W & 3) == 0;
    return c.wino && c.off_w4f >= 0 && wino4_min_nb() > 0 && NB >= wino4_min_nb() && (H & 3) == 0 && (W & 3) == 0 && !(ex.pack_skips & 16);
}
// F(4x4,3x3) for the stride-2 5x5 layers in phase form (wino43_kernels.hip) under the same conditions (OH x OW = the conv's output size)
static bool wino43_applies(const Exec& ex, const ConvSpec& c, int NB, int OH, int OW)
{
    if (ex.force_scheme == 1 || ex.force_scheme >= 3) return false;
    if (ex.force_scheme == 2) return c.wino3 && c.off_w43 >= 0 && (OH & 3) == 0 && (OW & 3) == 0;
    return c.wino3 && c.off_w43 >= 0 && wino43_min_nb() > 0 && NB >= wino43_min_nb() && (OH & 3) == 0 && (OW & 3) == 0 && !(ex.pack_skips & 32);
}
// samples per F(4x4) pass: V / M hold pts * max(K, M) * tiles floats
static int wino4_chunk(const Exec& ex, int NB, long long tiles_per_sample, int KM, int pts = 64)
{
    const long long cap_tiles = (ex.wino_cap / ((long long)pts * KM)) & ~31LL;
    if (tiles_per_sample <= 0 || cap_tiles < tiles_per_sample) return 0;
    const long long nbmax = cap_tiles / tiles_per_sample;
    if (NB <= nbmax) return NB;
    const long long nch = (NB + nbmax - 1) / nbmax;
    return (int)((NB + nch - 1) / nch);
}

static bool conv_wino4(Exec& ex, const ConvSpec& c, const float* packed, int dgrad, int NB, int H, int W, CView x, View y, int shuffle, int accumulate)
{
    const int K = dgrad ? c.cout_tot : c.Cin, M = dgrad ? c.Cin : c.cout_tot;
    const int TH = H / 4, TW = W / 4;
    if ((M % 128) != 0 || (K % 16) != 0) return false;
    const int nbc = wino4_chunk(ex, NB, (long long)TH * TW, K > M ? K : M);
    if (!nbc || ((long long)nbc * TH * TW < wino4_min_tiles() && ex.force_scheme != 2)) return false;
    if (ex.dry) return true;
    for (int b0 = 0; b0 < NB; b0 += nbc) {
        const int nb = NB - b0 < nbc ? NB - b0 : nbc;
        const long long NT = (long long)nb * TH * TW, NTp = NT <= 64 ? 64 : ((NT + 31) & ~31LL);      // (the GEMM wants a pitch of 64+ columns)
        WinoXformArgs xi{};
        xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv;
        xi.N = nb; xi.C = K; xi.H = H; xi.W = W; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 2;
        ex.fail(mcvc_wino4_input_launch(xi, ex.s));
        WinoGemmArgs ga{};
        ga.a = packed + (dgrad ? c.off_w4d : c.off_w4f); ga.a_xi = dgrad ? c.wd_xi : c.wf_xi; ga.lda = dgrad ? c.cin_pk : c.cout_pk;
        ga.b = ex.wv; ga.b_xi = (long long)K * NTp; ga.ldb = (int)NTp;
        ga.c = ex.wm; ga.c_xi = (long long)M * NTp; ga.ldc = (int)NTp;
        ga.M = M; ga.N = (int)NTp; ga.K = K; ga.nxi = 64;
        ex.fail(mcvc_wino_gemm_launch(ga, ex.s));
        WinoOutArgs oa{};
        oa.m = ex.wm; oa.bias = dgrad ? nullptr : packed + c.off_bias;
        oa.y = y.p + (long long)b0 * y.sb; oa.y_sb = y.sb; oa.y_sc = y.sc; oa.y_sh = y.sh;
        oa.N = nb; oa.Cout = M; oa.OH = H; oa.OW = W; oa.TH = TH; oa.TW = TW; oa.NT = (int)NT; oa.NTp = (int)NTp;
        oa.shuffle = shuffle; oa.YH = 2 * H; oa.YW = 2 * W; oa.accumulate = accumulate;
        if (ex.fuse_next && !dgrad && shuffle && nbc == NB) { ex.pend = oa; ex.pend_pts = 64; }
        else ex.fail(mcvc_wino4_output_launch(oa, ex.s));
    }
    return true;
}

static bool conv_wino(Exec& ex, const ConvSpec& c, const float* packed, int dgrad, int NB, int H, int W, CView x, View y, int shuffle,
                      int accumulate)
{
    if (!c.wino || !wino_enabled() || !ex.wv) return false;
    if (wino4_applies(ex, c, NB, H, W) && conv_wino4(ex, c, packed, dgrad, NB, H, W, x, y, shuffle, accumulate)) return true;
    // (bits 128 / 256: the 36-point sets of upSample1 / upSample2 were not refreshed -- gen_pack_cfg skips them for a layer whose EVERY pass at
    //  the engine's frame count takes the 4 x 4 scheme; recorded per LAYER, so a pass at another frame count on the same packed buffer that
    //  does reach this path reads no stale set: ADVICE r5)
    if (!ex.dry && (ex.pack_skips & (c.Cout == 1024 ? 128 : 256))) { ex.fail(MCVC_ERR_INVALID); return true; }
    const int K = dgrad ? c.cout_tot : c.Cin, M = dgrad ? c.Cin : c.cout_tot;
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const int nbc = wino_chunk(NB, (long long)TH * TW);
    if (!nbc || 36LL * (K > M ? K : M) * wino_chunk_tiles(NB, (long long)TH * TW) > ex.wino_cap) return false;
    if (ex.dry) return true;
    static const int own_gemm = mcvc_knob("MCVC_WINO_GEMM", 1);
    for (int b0 = 0; b0 < NB; b0 += nbc) {
        const int nb = NB - b0 < nbc ? NB - b0 : nbc;
        const long long NT = (long long)nb * TH * TW;
        // the 36 products see the tiles as a (NTp/32) x 32 "image" so that the conv kernel's 2-D pixel tiles are full
        const long long NTp = (NT + 31) & ~31LL;
        WinoXformArgs xi{};
        xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv;
        xi.N = nb; xi.C = K; xi.H = H; xi.W = W; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 2;
        ex.fail(mcvc_wino_input_launch(xi, ex.s));
        if (own_gemm && (M % 128) == 0 && (K % 16) == 0 && NTp >= 64) {
            WinoGemmArgs ga{};
            ga.a = packed + (dgrad ? c.off_wd : c.off_wf); ga.a_xi = dgrad ? c.wd_xi : c.wf_xi; ga.lda = dgrad ? c.cin_pk : c.cout_pk;
            ga.b = ex.wv; ga.b_xi = (long long)K * NTp; ga.ldb = (int)NTp;
            ga.c = ex.wm; ga.c_xi = (long long)M * NTp; ga.ldc = (int)NTp;
            ga.M = M; ga.N = (int)NTp; ga.K = K;
            ex.fail(mcvc_wino_gemm_launch(ga, ex.s));
        } else {
            ConvProblem p{K, (int)(NTp / 32), 32, M, (int)(NTp / 32), 32, 1, 1, 1, 0, 0};
            ConvIO io{};
            io.x = ex.wv; io.x_sb = (long long)K * NTp; io.x_sc = NTp; io.x_sh = 32;
            io.y = ex.wm; io.y_sb = (long long)M * NTp; io.y_sc = NTp; io.y_sh = 32; io.y_sw = 1;
            io.nsplit = 1;
            io.w_nstride = dgrad ? c.wd_xi : c.wf_xi;
            io.tile_cfg = wino_tile_cfg(M, NT);
            io.gemm_ok = 1;
            ex.fail(mcvc_conv_launch(p, 36, io, packed + (dgrad ? c.off_wd : c.off_wf), dgrad ? c.dg_rows_co : c.cin_pad, dgrad ? c.cin_pk : c.cout_pk,
                                     nullptr, ex.s, nullptr));
        }
        WinoOutArgs oa{};
        oa.m = ex.wm; oa.bias = dgrad ? nullptr : packed + c.off_bias;
        oa.y = y.p + (long long)b0 * y.sb; oa.y_sb = y.sb; oa.y_sc = y.sc; oa.y_sh = y.sh;
        oa.N = nb; oa.Cout = M; oa.OH = H; oa.OW = W; oa.TH = TH; oa.TW = TW; oa.NT = (int)NT; oa.NTp = (int)NTp;
        oa.shuffle = shuffle; oa.YH = 2 * H; oa.YW = 2 * W; oa.accumulate = accumulate;
        if (ex.fuse_next && !dgrad && shuffle && nbc == NB) { ex.pend = oa; ex.pend_pts = 36; }
        else ex.fail(mcvc_wino_output_launch(oa, ex.s));
    }
    return true;
}

// Convolutions that run as staged GEMMs on the LDS-DMA GEMM pipeline (sgemm.h): 1 x KW stride 1 (KW = 1, 3) over an image of rows -- the 1-D
// trunk beyond the fused small-batch kernels (more than 64 columns; MCVC_SGEMM1D_COLS = smallest column count, 0 = never).  (kind 1 of r2-r4,
// the discriminators' 3 x 3 stride-2 layers as staged products -- tap planes, transposed operands, gather -- is gone: those layers run as
// IMPLICIT GEMMs in every pass whose shapes allow it (igemm_applies) and on the direct kernels otherwise.)
struct SgKind { int kind, taps, OH, OW; };
static thread_local int t_no_sgemm = 0;
// (op-level entries: operands are the CALLER's tensors.  The 1 x 3 implicit forms read one float in front of / behind a tensor -- the shifted
//  windows' first / last piece, whose out-of-row element is then dropped -- which is inside the library's own stash / scratch layouts but not
//  guaranteed to be mapped for a foreign allocation: those entries keep the direct kernels for 1-D layers)
static thread_local int t_user_operands = 0;
static SgKind sgemm_kind(const ConvSpec& c, int NB, int H, int W)
{
    static const int min_cols = mcvc_knob("MCVC_SGEMM1D_COLS", 64);
    SgKind k{0, 0, 0, 0};
    if (t_no_sgemm || t_user_operands) return k;
    if (c.nbr > 2 || (c.cout_tot % 64) != 0 || (c.Cout % 32) != 0 || c.cin_pad != c.Cin) return k;
    if (c.KH == 1 && (c.KW == 1 || c.KW == 3) && c.stride == 1 && c.ph == 0 && c.pw == (c.KW - 1) / 2) {
        k.OH = H; k.OW = W; k.taps = c.KW;
        if (min_cols > 0 && (long long)NB * H * W >= min_cols && (c.Cin % 64) == 0 && (W & 3) == 0) k.kind = 2;
    }
    return k;
}
// K-split: the smallest power-of-two split (<= 8) that yields >= 256 workgroups while a slab keeps >= 64 k per half.  (A cost-model split
// against round quantisation -- stream-K's effect with uniform splits -- was built in r4 and measured SLOWER inside the concurrent lanes: bs=8
// 22.2-22.6 -> 23.0-23.2 ms, the other lanes' kernels already fill a product's last round; removed)
static int split_by_cost(long long tiles, int Kmin)
{
    static const int wgs = mcvc_knob("MCVC_SGEMM_WGS", 256);
    static const int maxsp = mcvc_knob("MCVC_SGEMM_MAXSPLIT", 8);
    int sp = 1;
    while (sp < maxsp && tiles * sp < wgs && (Kmin % (64 * sp)) == 0 && Kmin / (2 * sp) >= 64) sp *= 2;
    return sp;
}
static int sgemm_split(int M, long long N, int K)
{
    return split_by_cost((long long)(M / 64) * ((N + 63) / 64), K);
}

// ---- implicit GEMM for the 3x3 stride-2 padding-1 layers (sgemm.h): no tap planes, no gather kernel -------------------------------------
static bool igemm_enabled()
{
    static const int en = mcvc_knob("MCVC_IGEMM", 1);
    return en != 0;
}
static bool igemm_applies(const ConvSpec& c, int H, int W)
{
    return igemm_enabled() && c.igemm && (H & 1) == 0 && (W & 1) == 0 && ((W / 2) & 3) == 0 && H >= 2 && W >= 8;
}
// K split: enough 64 x 64 tiles x classes x splits to occupy the chip, at least two 32-deep stages per split of the shortest class
static int igemm_split(long long tiles, int Kmin) { return split_by_cost(tiles, Kmin); }
// forward: xs = the input in the phase-split padded layout, y = the conv output (dense planes); K-split slabs 1.. are summed by the consumer
static void conv_fwd_igemm(Exec& ex, const ConvSpec& c, const float* packed, int NB, int H, int W, const float* xs, View y, long long y_total,
                           int allow_split, int* nsplit)
{
    const int OH = H / 2, OW = W / 2, P = OH * OW, KT = 9 * c.Cin;
    const long long NT = (long long)NB * P;
    const int sp = (allow_split && nsplit) ? igemm_split((long long)(c.cout_tot / 64) * ((NT + 63) / 64), KT) : 1;
    const long long slab_need = (long long)(sp - 1) * y_total;
    if (slab_need > ex.slab_need) ex.slab_need = slab_need;
    if (nsplit) *nsplit = sp;
    if (ex.dry) return;
    if (slab_need > ex.slab_cap) { ex.fail(MCVC_ERR_WORKSPACE); return; }
    const int pw = mcvc_xs_pw(W);
    const long long plane = mcvc_xs_plane(H, W);
    IGemmArgs g{};
    g.ncls = 1;
    g.cls[0].a = packed + c.off_ifwd; g.cls[0].ntaps = 9; g.cls[0].coff = 0;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {          // tap (kh, kw) of output (oh, ow) reads x[2 oh + kh - 1][2 ow + kw - 1]
            const int ph = (kh == 1) ? 0 : 1, di = (kh == 0) ? -1 : 0, pq = (kw == 1) ? 0 : 1, dj = (kw == 0) ? -1 : 0;
            g.cls[0].boff[3 * kh + kw] = (long long)(2 * ph + pq) * plane + (long long)(1 + di) * pw + 4 + dj;
            g.cls[0].aoff[3 * kh + kw] = (long long)(3 * kh + kw) * c.Cin * c.cout_pk;      // tap-major copy: rows (tap, ci)
        }
    g.a_ks = c.cout_pk;
    g.b = xs; g.b_cs = 4 * plane; g.b_sn = (long long)c.Cin * 4 * plane; g.b_pitch = pw; g.Cb = c.Cin; g.OW = OW; g.P = P;
    g.c = y.p; g.ldc = y.sc; g.c_sn = y.sb; g.c_sh = y.sh; g.c_sw = 1;
    g.bias = packed + c.off_bias;
    g.M = c.cout_tot; g.N = (int)NT; g.nsplit = sp; g.c_slab = ex.slabs; g.c_split = y_total;
    ex.fail(mcvc_igemm_launch(g, ex.s));
}
// data gradient: dyp = dY in the padded layout (planes of (OH + 1) x (OW + 4), zero borders), dx dense; the four output-parity classes in
// one launch, every input pixel written by exactly one of them; K-split slabs (dx-shaped) are summed by the consumer
static void conv_dgrad_igemm(Exec& ex, const ConvSpec& c, const float* packed, int NB, int H, int W, const float* dyp, View dx, long long dx_total,
                             int accumulate, int allow_split, int* nsplit)
{
    const int OH = H / 2, OW = W / 2, P = OH * OW;
    const long long NT = (long long)NB * P;
    int sp = (allow_split && nsplit && !accumulate) ? igemm_split(4LL * (c.Cin / 64) * ((NT + 63) / 64), c.cout_tot) : 1;
    // (Tried on top, r5: deeper uniform splits -- 4-tap class at most 512 deep -- and tap-wise splits with zero-filled slabs for the classes with
    //  fewer taps, so that every workgroup is one tap deep: bs=1 6.01 -> 6.03-6.12 and -> 6.3 ms.  The consumer reading 4-8 slabs costs more
    //  than the imbalance of the classes; the uniform split of r4 stays.)
    const long long slab_need = (long long)(sp - 1) * dx_total;
    if (slab_need > ex.slab_need) ex.slab_need = slab_need;
    if (nsplit) *nsplit = sp;
    if (ex.dry) return;
    if (slab_need > ex.slab_cap) { ex.fail(MCVC_ERR_WORKSPACE); return; }
    const int pitch = mcvc_dyp_pitch(OW);
    const long long plane = mcvc_dyp_plane(OH, OW);
    IGemmArgs g{};
    g.ncls = c.ncls;
    for (int k = 0; k < c.ncls; ++k) {
        const DgradClass& d = c.icls[k];
        IGemmClass& q = g.cls[k];
        q.a = packed + c.off_ifwd; q.ntaps = d.nth * d.ntw;          // the tap-major FORWARD copy, read row-major (sgemm.h arow)
        q.coff = (long long)d.qh * dx.sh + d.qw;
        for (int u = 0; u < d.nth; ++u)
            for (int v = 0; v < d.ntw; ++v) {     // input row 2a + qh receives tap kh from output row a + (qh + 1 - kh) / 2
                const int kh = d.khmax - 2 * u, kw = d.kwmax - 2 * v;
                q.boff[u * d.ntw + v] = (long long)((d.qh + 1 - kh) / 2) * pitch + (d.qw + 1 - kw) / 2;
                q.aoff[u * d.ntw + v] = (long long)(3 * kh + kw) * c.Cin * c.cout_pk;
            }
    }
    g.a_ks = c.cout_pk; g.arow = 1;
    g.b = dyp; g.b_cs = plane; g.b_sn = (long long)c.cout_tot * plane; g.b_pitch = pitch; g.Cb = c.cout_tot; g.OW = OW; g.P = P;
    g.c = dx.p; g.ldc = dx.sc; g.c_sn = dx.sb; g.c_sh = 2 * dx.sh; g.c_sw = 2; g.accumulate = accumulate;
    g.M = c.Cin; g.N = (int)NT; g.nsplit = sp; g.c_slab = ex.slabs; g.c_split = dx_total;
    ex.fail(mcvc_igemm_launch(g, ex.s));
}

static void conv_fwd(Exec& ex, const ConvSpec& c, const float* packed, int NB, int H, int W, CView x, View y, long long y_total,
                     int shuffle, int allow_split, int* nsplit)
{
    const SgKind sk = shuffle ? SgKind{0, 0, 0, 0} : sgemm_kind(c, NB, H, W);
    if (sk.kind) {
        const int P = sk.OH * sk.OW, KT = sk.taps * c.Cin;
        const long long NT = (long long)NB * P;
        const bool b_in_place = c.KW == 1;                                                    // a 1x1 conv multiplies x itself
        const int sp = (allow_split && nsplit) ? sgemm_split(c.cout_tot, NT, KT) : 1;       // slabs 1.. are summed by the consumer (norm / act)
        const long long slab_need = (long long)(sp - 1) * y_total;
        if (ex.dry) { if (slab_need > ex.slab_need) ex.slab_need = slab_need; if (nsplit) *nsplit = sp; return; }
        if ((ex.pack_skips & 1) && c.off_tk >= 0) { ex.fail(MCVC_ERR_INVALID); return; }     // stale K-major copy
        if (!b_in_place && slab_need <= ex.slab_cap && y.sh == sk.OW && y.sc == P && (NB == 1 || y.sb == (long long)c.cout_tot * P)) {
            // 1 x 3 over dense rows: an implicit GEMM (sgemm.h) -- the three windows of x are gathered where they lie, shifted by -1 / 0 / +1
            // column; the K-major forward copy's rows are (ci, kw), so a tap's rows are KW apart
            IGemmArgs g{};
            g.ncls = 1;
            IGemmClass& q = g.cls[0];
            q.a = packed + c.off_fwd; q.ntaps = c.KW; q.coff = 0;
            for (int kw = 0; kw < c.KW; ++kw) { q.boff[kw] = kw - c.pw; q.aoff[kw] = (long long)kw * c.cout_pk; q.zs[kw] = kw < c.pw ? 1 : (kw > c.pw ? 2 : 0); }
            g.a_ks = (long long)c.KW * c.cout_pk; g.zw = W;
            g.b = x.p; g.b_cs = x.sc; g.b_sn = x.sb; g.b_pitch = x.sh; g.Cb = c.Cin; g.OW = W; g.P = P;
            g.c = y.p; g.ldc = y.sc; g.c_sn = y.sb; g.c_sh = y.sh; g.c_sw = 1;
            g.bias = packed + c.off_bias;
            g.M = c.cout_tot; g.N = (int)NT; g.nsplit = sp; g.c_slab = ex.slabs; g.c_split = y_total;
            ex.fail(mcvc_igemm_launch(g, ex.s));
            if (nsplit) *nsplit = sp;
            return;
        }
        if (b_in_place && x.sh == W && x.sc == P && slab_need <= ex.slab_cap && y.sh == sk.OW && y.sc == P &&
            (NB == 1 || y.sb == (long long)c.cout_tot * P)) {
            SGemmArgs g{};
            g.a = packed + c.off_fwd; g.lda = c.cout_pk;                       // Wt[k = ci][co] (value | gate columns)
            g.b = x.p; g.ldb = x.sc; g.bseg = P; g.b_sn = x.sb;
            g.c = y.p; g.ldc = y.sc; g.cseg = P; g.c_sn = y.sb;
            g.bias = packed + c.off_bias;
            g.M = c.cout_tot; g.N = (int)NT; g.K = KT; g.nsplit = sp; g.c_slab = ex.slabs; g.c_split = y_total;
            ex.fail(mcvc_sgemm_launch(g, ex.s));
            if (nsplit) *nsplit = sp;
            return;
        }
    }
    if (conv_wino(ex, c, packed, 0, NB, H, W, x, y, shuffle, 0)) { if (nsplit) *nsplit = 1; return; }
    if (c.wino3 && wino_enabled() && ex.wv && !shuffle && (H & 1) == 0 && (W & 1) == 0 && (c.cout_tot % 128) == 0 && wino43_applies(ex, c, NB, H / 2, W / 2)) {
        // ... as F(4x4,3x3): 36 points per 16 outputs
        static const int en = mcvc_knob("MCVC_WINO3_FWD", 1);
        const int OH = H / 2, OW = W / 2, K = 4 * c.Cin, M = c.cout_tot;
        const int TH = OH / 4, TW = OW / 4;
        const int nbc = wino4_chunk(ex, NB, (long long)TH * TW, K > M ? K : M, 36);
        if (en && nbc && ((long long)nbc * TH * TW >= 64 || ex.force_scheme == 2)) {
            if (nsplit) *nsplit = 1;
            if (ex.dry) return;
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv;
                xi.N = nb; xi.C = K; xi.H = OH; xi.W = OW; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 1;
                ex.fail(mcvc_wino43_input_phase_launch(xi, H, W, ex.s));
                WinoGemmArgs ga{};
                ga.a = packed + c.off_w43f; ga.a_xi = c.w3f_xi; ga.lda = c.cout_pk;
                ga.b = ex.wv; ga.b_xi = (long long)K * NTp; ga.ldb = (int)NTp;
                ga.c = ex.wm; ga.c_xi = (long long)M * NTp; ga.ldc = (int)NTp;
                ga.M = M; ga.N = (int)NTp; ga.K = K; ga.nxi = 36;
                ex.fail(mcvc_wino_gemm_launch(ga, ex.s));
                WinoOutArgs oa{};
                oa.m = ex.wm; oa.bias = packed + c.off_bias;
                oa.y = y.p + (long long)b0 * y.sb; oa.y_sb = y.sb; oa.y_sc = y.sc; oa.y_sh = y.sh;
                oa.N = nb; oa.Cout = M; oa.OH = OH; oa.OW = OW; oa.TH = TH; oa.TW = TW; oa.NT = (int)NT; oa.NTp = (int)NTp;
                oa.shuffle = 0; oa.YH = OH; oa.YW = OW; oa.accumulate = 0;
                if (ex.fuse_next && nbc == NB) { ex.pend = oa; ex.pend_pts = 43; }
                else ex.fail(mcvc_wino43_output_launch(oa, ex.s));
            }
            return;
        }
    }
    if (c.wino3 && wino_enabled() && ex.wv && !shuffle && (H & 1) == 0 && (W & 1) == 0 && (c.cout_tot % 128) == 0) {
        // stride-2 5x5 forward = 3x3 stride-1 conv over the four input phases: Winograd F(2x2,3x3), K = 4*Cin
        static const int en = mcvc_knob("MCVC_WINO3_FWD", 1);
        const int OH = H / 2, OW = W / 2, K = 4 * c.Cin, M = c.cout_tot;
        const int TH = (OH + 1) / 2, TW = (OW + 1) / 2;
        const int nbc = wino_chunk(NB, (long long)TH * TW);
        const long long NTc = wino_chunk_tiles(NB, (long long)TH * TW);
        if (en && nbc && NTc >= 64 && 16LL * (K > M ? K : M) * NTc <= ex.wino_cap) {
            if (nsplit) *nsplit = 1;
            if (ex.dry) return;
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv;
                xi.N = nb; xi.C = K; xi.H = OH; xi.W = OW; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 1;
                ex.fail(mcvc_wino3_input_phase_launch(xi, H, W, ex.s));
                WinoGemmArgs ga{};
                ga.a = packed + c.off_w3f; ga.a_xi = c.w3f_xi; ga.lda = c.cout_pk;
                ga.b = ex.wv; ga.b_xi = (long long)K * NTp; ga.ldb = (int)NTp;
                ga.c = ex.wm; ga.c_xi = (long long)M * NTp; ga.ldc = (int)NTp;
                ga.M = M; ga.N = (int)NTp; ga.K = K; ga.nxi = 16;
                ex.fail(mcvc_wino_gemm_launch(ga, ex.s));
                WinoOutArgs oa{};
                oa.m = ex.wm; oa.bias = packed + c.off_bias;
                oa.y = y.p + (long long)b0 * y.sb; oa.y_sb = y.sb; oa.y_sc = y.sc; oa.y_sh = y.sh;
                oa.N = nb; oa.Cout = M; oa.OH = OH; oa.OW = OW; oa.TH = TH; oa.TW = TW; oa.NT = (int)NT; oa.NTp = (int)NTp;
                oa.shuffle = 0; oa.YH = OH; oa.YW = OW; oa.accumulate = 0;
                if (ex.fuse_next && nbc == NB) { ex.pend = oa; ex.pend_pts = 16; }
                else ex.fail(mcvc_wino3_output_launch(oa, ex.s));
            }
            return;
        }
    }
    if (!ex.dry && (((ex.pack_skips & 1) && c.off_tk >= 0) || ((ex.pack_skips & 2) && (c.wino || c.wino3)))) { ex.fail(MCVC_ERR_INVALID); return; }
    ConvProblem p{c.Cin, H, W, c.cout_tot, conv_out(H, c.KH, c.stride, c.ph), conv_out(W, c.KW, c.stride, c.pw),
                  c.KH, c.KW, c.stride, c.ph, c.pw};
    ConvIO io{};
    io.x = x.p; io.x_sb = x.sb; io.x_sc = x.sc; io.x_sh = x.sh;
    io.y = y.p; io.y_sb = y.sb; io.y_sc = y.sc; io.y_sh = y.sh; io.y_sw = 1;
    io.shuffle = shuffle;
    run_conv(ex, p, NB, io, y_total, packed + c.off_fwd, c.w_rows, c.cout_pk, packed + c.off_bias, allow_split, 0, nsplit);
}

// dX = conv-backward-data.  dy: conv-output layout (cout_tot channels, OHxOW); dx: input layout.
static void conv_dgrad(Exec& ex, const ConvSpec& c, const float* packed, int NB, int H, int W, CView dy, View dx, long long dx_total,
                       int accumulate, int allow_split, int* nsplit)
{
    if (conv_wino(ex, c, packed, 1, NB, H, W, dy, dx, 0, accumulate)) { if (nsplit) *nsplit = 1; return; }
    const int OH = conv_out(H, c.KH, c.stride, c.ph), OW = conv_out(W, c.KW, c.stride, c.pw);
    const int st = c.stride;
    const SgKind sk = sgemm_kind(c, NB, H, W);
    if (sk.kind) {
        // 1 x KW over dense rows as an implicit GEMM (sgemm.h): M = input channels, K = (kw, output channel); A = the K-major FORWARD copy read
        // row-major (rows (ci, kw), output channels contiguous: arow), B = the windows of dY shifted by pw - kw columns.  An accumulating
        // destination with a K split leaves dx alone: all splits go to slabs and the consumer sums dx + slabs (*nsplit = splits + 1).
        const int P = sk.OH * sk.OW;
        const long long NT = (long long)NB * P;
        int sp = (allow_split && nsplit) ? igemm_split((long long)(c.Cin / 64) * ((NT + 63) / 64), sk.taps * c.cout_tot) : 1;
        while (sp > 1 && ((long long)sk.taps * c.cout_tot) % (32LL * sp) != 0) sp /= 2;
        const bool slab_all = accumulate && sp > 1;
        const long long slab_need = (long long)(slab_all ? sp : sp - 1) * dx_total;
        if (slab_need > ex.slab_need) ex.slab_need = slab_need;
        if (nsplit) *nsplit = slab_all ? sp + 1 : sp;
        if (ex.dry) return;
        if ((ex.pack_skips & 1) && c.off_tk >= 0) { ex.fail(MCVC_ERR_INVALID); return; }     // stale K-major copy
        if (slab_need > ex.slab_cap) { ex.fail(MCVC_ERR_WORKSPACE); return; }
        IGemmArgs g{};
        g.ncls = 1;
        IGemmClass& q = g.cls[0];
        q.a = packed + c.off_fwd; q.ntaps = c.KW; q.coff = 0;
        for (int kw = 0; kw < c.KW; ++kw) {
            const int sh = c.pw - kw;                 // dX[n] += W[:, :, kw]^T dY[n + pw - kw]
            q.boff[kw] = sh; q.aoff[kw] = (long long)kw * c.cout_pk; q.zs[kw] = sh < 0 ? 1 : (sh > 0 ? 2 : 0);
        }
        g.a_ks = (long long)c.KW * c.cout_pk; g.arow = 1; g.zw = W;
        g.b = dy.p; g.b_cs = dy.sc; g.b_sn = dy.sb; g.b_pitch = dy.sh; g.Cb = c.cout_tot; g.OW = W; g.P = P;
        g.ldc = dx.sc; g.c_sn = dx.sb; g.c_sh = dx.sh; g.c_sw = 1;
        g.M = c.Cin; g.N = (int)NT; g.nsplit = sp; g.c_split = dx_total;
        if (slab_all) { g.c = ex.slabs; g.c_slab = ex.slabs + dx_total; }
        else { g.c = dx.p; g.c_slab = ex.slabs; g.accumulate = accumulate; }
        ex.fail(mcvc_igemm_launch(g, ex.s));
        return;
    }
    if (c.wino3 && wino_enabled() && ex.wv && wino43_applies(ex, c, NB, OH, OW) && 2 * OH == H && 2 * OW == W) {
        // ... as F(4x4,3x3)
        static const int en = mcvc_knob("MCVC_WINO3", 1);
        const int K = c.cout_tot, M = 4 * c.Cin;
        const int TH = OH / 4, TW = OW / 4;
        const int nbc = wino4_chunk(ex, NB, (long long)TH * TW, K > M ? K : M, 36);
        if (en && nbc && ((long long)nbc * TH * TW >= 64 || ex.force_scheme == 2) && (M % 128) == 0 && (K % 16) == 0) {
            if (nsplit) *nsplit = 1;
            if (ex.dry) return;
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = dy.p + (long long)b0 * dy.sb; xi.x_sb = dy.sb; xi.x_sc = dy.sc; xi.x_sh = dy.sh; xi.v = ex.wv;
                xi.N = nb; xi.C = K; xi.H = OH; xi.W = OW; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 1;
                ex.fail(mcvc_wino43_input_launch(xi, ex.s));
                WinoGemmArgs ga{};
                ga.a = packed + c.off_w43; ga.a_xi = c.w3_xi; ga.lda = c.mg_ld;
                ga.b = ex.wv; ga.b_xi = (long long)K * NTp; ga.ldb = (int)NTp;
                ga.c = ex.wm; ga.c_xi = (long long)M * NTp; ga.ldc = (int)NTp;
                ga.M = M; ga.N = (int)NTp; ga.K = K; ga.nxi = 36;
                ex.fail(mcvc_wino_gemm_launch(ga, ex.s));
                WinoOutArgs oa{};
                oa.m = ex.wm; oa.bias = nullptr;
                oa.y = dx.p + (long long)b0 * dx.sb; oa.y_sb = dx.sb; oa.y_sc = dx.sc; oa.y_sh = dx.sh;
                oa.N = nb; oa.Cout = M; oa.OH = OH; oa.OW = OW; oa.TH = TH; oa.TW = TW; oa.NT = (int)NT; oa.NTp = (int)NTp;
                oa.shuffle = 1; oa.YH = H; oa.YW = W; oa.accumulate = accumulate;
                ex.fail(mcvc_wino43_output_launch(oa, ex.s));
            }
            return;
        }
    }
    if (c.wino3 && wino_enabled() && ex.wv) {
        // merged stride-2 data-gradient = a 3x3 stride-1 conv over dY with 4*Cin output channels: Winograd F(2x2,3x3)
        static const int en = mcvc_knob("MCVC_WINO3", 1);
        const int K = c.cout_tot, M = 4 * c.Cin;
        const int TH = (OH + 1) / 2, TW = (OW + 1) / 2;
        const int nbc = wino_chunk(NB, (long long)TH * TW);
        const long long NTc = wino_chunk_tiles(NB, (long long)TH * TW);
        if (en && nbc && NTc >= 64 && 16LL * (K > M ? K : M) * NTc <= ex.wino_cap && (H + 1) / 2 == OH && (W + 1) / 2 == OW) {
            if (nsplit) *nsplit = 1;
            if (ex.dry) return;
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = dy.p + (long long)b0 * dy.sb; xi.x_sb = dy.sb; xi.x_sc = dy.sc; xi.x_sh = dy.sh; xi.v = ex.wv;
                xi.N = nb; xi.C = K; xi.H = OH; xi.W = OW; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 1;
                ex.fail(mcvc_wino3_input_launch(xi, ex.s));
                WinoGemmArgs ga{};
                ga.a = packed + c.off_w3; ga.a_xi = c.w3_xi; ga.lda = c.mg_ld;
                ga.b = ex.wv; ga.b_xi = (long long)K * NTp; ga.ldb = (int)NTp;
                ga.c = ex.wm; ga.c_xi = (long long)M * NTp; ga.ldc = (int)NTp;
                ga.M = M; ga.N = (int)NTp; ga.K = K; ga.nxi = 16;
                ex.fail(mcvc_wino_gemm_launch(ga, ex.s));
                WinoOutArgs oa{};
                oa.m = ex.wm; oa.bias = nullptr;
                oa.y = dx.p + (long long)b0 * dx.sb; oa.y_sb = dx.sb; oa.y_sc = dx.sc; oa.y_sh = dx.sh;
                oa.N = nb; oa.Cout = M; oa.OH = OH; oa.OW = OW; oa.TH = TH; oa.TW = TW; oa.NT = (int)NT; oa.NTp = (int)NTp;
                oa.shuffle = 1; oa.YH = H; oa.YW = W; oa.accumulate = accumulate;
                ex.fail(mcvc_wino3_output_launch(oa, ex.s));
            }
            return;
        }
    }
    if (!ex.dry && (((ex.pack_skips & 1) && c.off_tk >= 0) || ((ex.pack_skips & 2) && (c.wino || c.wino3)))) { ex.fail(MCVC_ERR_INVALID); return; }
    if (!ex.dry && (ex.pack_skips & 8) && c.stride == 2) { ex.fail(MCVC_ERR_INVALID); return; }      // (mcvc_disc_pack_small: these copies are stale)
    static const int ucls_nb = mcvc_knob("MCVC_DGRAD_CLASSES_NB", 8);
    const bool per_class = c.merged && c.off_dcls >= 0 && NB >= ucls_nb;
    if (c.merged && !per_class) {
        ConvProblem p{c.cout_tot, OH, OW, 4 * c.Cin, (H + 1) / 2, (W + 1) / 2, c.mg_kh, c.mg_kw, 1, c.mg_pad_h, c.mg_pad_w};
        ConvIO io{};
        io.x = dy.p; io.x_sb = dy.sb; io.x_sc = dy.sc; io.x_sh = dy.sh;
        io.y = dx.p; io.y_sb = dx.sb; io.y_sc = dx.sc; io.y_sh = dx.sh; io.y_sw = 1;
        io.accumulate = accumulate; io.shuffle = 1; io.YH = H; io.YW = W;
        run_conv(ex, p, NB, io, dx_total, packed + c.off_dgrad, c.dg_rows_co * c.mg_kh * c.mg_kw, c.mg_ld, nullptr, allow_split, 0, nsplit);
        return;
    }
    const DgradClass* klass = per_class ? c.ucls : c.cls;
    const long long kbase = per_class ? c.off_dcls : c.off_dgrad;
    ConvProblem ps[4];
    int force = 0;
    for (int k = 0; k < c.ncls; ++k) {
        const DgradClass& d = klass[k];
        ps[k] = ConvProblem{c.cout_tot, OH, OW, c.Cin, (H - d.qh + st - 1) / st, (W - d.qw + st - 1) / st,
                            d.nth, d.ntw, 1, d.pad_h, d.pad_w};
    }
    if (c.ncls > 1 && allow_split) {          // all parity classes must agree on the slab count
        force = 1 << 30;
        for (int k = 0; k < c.ncls; ++k) { const int n = mcvc_conv_plan_nsplit(ps[k], NB, 1); if (n < force) force = n; }
        if (force < 1) force = 1;
        if (ex.max_split > 0 && force > ex.max_split) force = ex.max_split;
    }
    int ns_all = 1;
    for (int k = 0; k < c.ncls; ++k) {
        const DgradClass& d = klass[k];
        if (ps[k].OH <= 0 || ps[k].OW <= 0) continue;
        ConvIO io{};
        io.x = dy.p; io.x_sb = dy.sb; io.x_sc = dy.sc; io.x_sh = dy.sh;
        io.y = dx.p + (long long)d.qh * dx.sh + d.qw; io.y_sb = dx.sb; io.y_sc = dx.sc; io.y_sh = dx.sh * st; io.y_sw = st;
        io.accumulate = accumulate;
        int ns = 1;
        // slabs of class k live at the same slab base + the class's element offset
        Exec sub = ex;
        if (!ex.dry) sub.slabs = ex.slabs + (long long)d.qh * dx.sh + d.qw;
        run_conv(sub, ps[k], NB, io, dx_total, packed + kbase + d.offset, c.dg_rows_co * d.nth * d.ntw, c.cin_pk, nullptr,
                 allow_split, force, &ns);
        ex.err = sub.err; ex.slab_need = sub.slab_need;
        if (ns > ns_all) ns_all = ns;
    }
    if (nsplit) *nsplit = ns_all;
}

static int wgrad_cin1_enabled()
{
    static const int en = mcvc_knob("MCVC_WGRAD_CIN1", 1);
    return en;
}

static void conv_wgrad(Exec& ex, const ConvSpec& c, float* const* grads, int NB, int H, int W, CView x, CView dy)
{
    const int OH = conv_out(H, c.KH, c.stride, c.ph), OW = conv_out(W, c.KW, c.stride, c.pw);
    ConvProblem p{c.Cin, H, W, c.Cout, OH, OW, c.KH, c.KW, c.stride, c.ph, c.pw};
    // GEMM form (sgemm.h): pixel-major operands in the weight gradients' own staging region, K-split slabs, then dw += slabs
    const SgKind sk = sgemm_kind(c, NB, H, W);                            // (the 1 x KW layers of the wide trunk)
    const bool ig3 = ex.wgrad_x_xs && igemm_applies(c, H, W);             // a 3 x 3 stride-2 layer whose input arrives phase-split
    const int wtaps = ig3 ? 9 : sk.taps;
    const int KT = wtaps * c.Cin;
    long long sg_floats = 0;
    // implicit form (r5, wgemm_kernels.hip): x arrives phase-split (the layer's forward ran as an implicit GEMM) -- both operands are read where
    // they lie; K split over the pixels when 128 x 32-filter tiles alone cannot fill the chip (slabs in the same staging region, dw_accum sums)
    static const int wgemm_on = mcvc_knob("MCVC_WGEMM", 1);
    const bool implicit = wgemm_on && (ig3 || sk.kind == 2) && (c.cout_tot % 128) == 0 && (c.Cin % mcvc_wgemm_cib(wtaps)) == 0 &&
                          (c.nbr == 1 || (c.Cout % 32) == 0);
    int wg_split = 1;
    if (implicit) {
        const int tiles = (c.cout_tot / 128) * (c.Cin / mcvc_wgemm_cib(wtaps));
        const long long nst = ((long long)NB * OH * OW + 31) / 32;
        static const int wg_target = mcvc_knob("MCVC_WGEMM_WGS", 256), wg_max = mcvc_knob("MCVC_WGEMM_MAXSPLIT", 64);
        // (a workgroup per compute unit -- a tile holds 156 KB of LDS -- and at least two pixel stages per split)
        while (wg_split < wg_max && tiles * wg_split < wg_target && nst / (2 * wg_split) >= 2) wg_split *= 2;
        sg_floats = wg_split > 1 ? (long long)wg_split * c.cout_tot * KT : 0;
    }
    if (ex.dry) {          // K-split slabs: their own region, so they never alias the data-gradient slabs of the main stream
        if (implicit && sg_floats > ex.sgw_need) ex.sgw_need = sg_floats;          // (the direct kernel's slabs stay reserved as the fallback)
        const long long need = mcvc_wgrad_plan_slab_floats(p, NB);
        if (need > ex.wslab_need) ex.wslab_need = need;
        return;
    }
    if (!grads) return;
    hipStream_t ws = ex.s;
    if (ex.s2) {           // dY (and x) are complete on the main stream at this point
        hipEvent_t e = pool_event();
        ex.fail(mcvc_event_record(e, ex.s));
        ex.fail(mcvc_stream_wait(ex.s2, e));
        ws = ex.s2;
    }
    bool done = false;
    if (implicit && (wg_split == 1 || (ex.sgw && sg_floats <= ex.sgw_cap)) && grads[c.wi[0]] && (c.nbr == 1 || grads[c.wi[1]])) {
        WGemmArgs g{};
        g.a = dy.p; g.a_cs = dy.sc; g.a_sn = dy.sb; g.a_pitch = dy.sh;
        g.b = x.p; g.b_cs = x.sc; g.b_sn = x.sb; g.ntaps = wtaps;
        if (ig3) {
            const int pw = mcvc_xs_pw(W);
            const long long plane = mcvc_xs_plane(H, W);
            g.b_pitch = pw;
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {          // tap (kh, kw) of output (oh, ow) multiplies x[2 oh + kh - 1][2 ow + kw - 1]  (conv_fwd_igemm)
                    const int ph = (kh == 1) ? 0 : 1, di = (kh == 0) ? -1 : 0, pq = (kw == 1) ? 0 : 1, dj = (kw == 0) ? -1 : 0;
                    g.boff[3 * kh + kw] = (long long)(2 * ph + pq) * plane + (long long)(1 + di) * pw + 4 + dj;
                }
        } else {                                          // 1 x KW over dense rows: tap kw multiplies x[w + kw - pw]; the rows' ends count as padding
            g.b_pitch = x.sh; g.zw = (c.KW == 3) ? W : 0;
            for (int kw = 0; kw < c.KW; ++kw) g.boff[kw] = kw - c.pw;
        }
        g.OW = OW; g.P = OH * OW; g.NPIX = NB * g.P;
        g.M = c.cout_tot; g.Cin = c.Cin; g.nsplit = wg_split;
        if (wg_split == 1) {
            g.c = grads[c.wi[0]]; g.accumulate = 1;
            if (c.nbr == 2) { g.c2 = grads[c.wi[1]]; g.m_split = c.Cout; }
            ex.fail(mcvc_wgemm_launch(g, ws));
        } else {
            g.c = ex.sgw; g.c_split = (long long)c.cout_tot * KT; g.c_slab = ex.sgw + g.c_split;
            ex.fail(mcvc_wgemm_launch(g, ws));
            ex.fail(mcvc_dw_accum_launch(ex.sgw, wg_split, g.c_split, grads[c.wi[0]], c.nbr == 2 ? grads[c.wi[1]] : nullptr, c.Cout, c.cout_tot, KT, ws));
        }
        done = true;
    }
    if (!done && ex.wgrad_x_xs) { ex.fail(MCVC_ERR_WORKSPACE); return; }     // (nothing else reads a phase-split input: never fall through to a dense reader)
    if (!done && wino_enabled() && ex.wu && c.nbr == 1 && grads[c.wi[0]] && wino4_applies(ex, c, NB, H, W) && (c.Cout % 128) == 0 && (c.Cin % 64) == 0 &&
        64LL * c.Cout * c.Cin <= ex.wu_cap) {
        // F(4x4,5x5) weight gradient: the same three steps on 8x8 tiles and 64 points
        const int TH = H / 4, TW = W / 4;
        const int nbc = wino4_chunk(ex, NB, (long long)TH * TW, c.Cout > c.Cin ? c.Cout : c.Cin);
        if (nbc && ((long long)nbc * TH * TW >= 32 || ex.force_scheme == 2)) {
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv2;
                xi.N = nb; xi.C = c.Cin; xi.H = H; xi.W = W; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 2;
                ex.fail(mcvc_wino4_input_t_launch(xi, ws));
                WinoXformArgs di{};
                di.x = dy.p + (long long)b0 * dy.sb; di.x_sb = dy.sb; di.x_sc = dy.sc; di.x_sh = dy.sh; di.v = ex.wm2;
                di.N = nb; di.C = c.Cout; di.H = H; di.W = W; di.TH = TH; di.TW = TW; di.NT = (int)NT; di.NTp = (int)NTp; di.pad = 0;
                ex.fail(mcvc_wino4_dy_t_launch(di, ws));
                WinoGemmArgs ga{};
                ga.a = ex.wm2; ga.a_xi = NTp * c.Cout; ga.lda = c.Cout;
                ga.b = ex.wv2; ga.b_xi = NTp * c.Cin; ga.ldb = c.Cin;
                ga.c = ex.wu; ga.c_xi = (long long)c.Cout * c.Cin; ga.ldc = c.Cin;
                ga.M = c.Cout; ga.N = c.Cin; ga.K = (int)((NT + 15) & ~15LL);      /* (rows [NT, NTp) of the tile-major operands are zeros: contract over the 16-row stages that hold tiles -- 80 instead of 96 at one sample) */ ga.nxi = 64;
                ex.fail(mcvc_wino_gemm_launch(ga, ws));
                ex.fail(mcvc_wino4_dw_launch(ex.wu, grads[c.wi[0]], c.Cout, c.Cin, ws));
            }
            done = true;
        }
    }
    if (!done && c.wino && wino_enabled() && ex.wu && c.nbr == 1 && grads[c.wi[0]]) {
        // Winograd weight gradient: dU[xi] = dM[xi] V[xi]^T over the tiles, then dW += G^T dU G.  Operands tile-major.
        static const int en = mcvc_knob("MCVC_WINO_WGRAD", 1);
        const int TH = (H + 1) / 2, TW = (W + 1) / 2;
        const int nbc = wino_chunk(NB, (long long)TH * TW);
        const long long NTc = wino_chunk_tiles(NB, (long long)TH * TW);
        if (en && nbc && (c.Cout % 128) == 0 && (c.Cin % 64) == 0 && 36LL * NTc * c.Cout <= ex.wino_cap && 36LL * NTc * c.Cin <= ex.wino_cap &&
            36LL * c.Cout * c.Cin <= ex.wu_cap) {
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv2;
                xi.N = nb; xi.C = c.Cin; xi.H = H; xi.W = W; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 2;
                ex.fail(mcvc_wino_input_t_launch(xi, ws));
                WinoXformArgs di{};
                di.x = dy.p + (long long)b0 * dy.sb; di.x_sb = dy.sb; di.x_sc = dy.sc; di.x_sh = dy.sh; di.v = ex.wm2;
                di.N = nb; di.C = c.Cout; di.H = H; di.W = W; di.TH = TH; di.TW = TW; di.NT = (int)NT; di.NTp = (int)NTp; di.pad = 0;
                ex.fail(mcvc_wino_dy_t_launch(di, ws));
                WinoGemmArgs ga{};
                ga.a = ex.wm2; ga.a_xi = NTp * c.Cout; ga.lda = c.Cout;          // dMt[xi][tile][co]
                ga.b = ex.wv2; ga.b_xi = NTp * c.Cin; ga.ldb = c.Cin;            // Vt[xi][tile][ci]
                ga.c = ex.wu; ga.c_xi = (long long)c.Cout * c.Cin; ga.ldc = c.Cin;
                ga.M = c.Cout; ga.N = c.Cin; ga.K = (int)((NT + 15) & ~15LL);      /* (rows [NT, NTp) of the tile-major operands are zeros: contract over the 16-row stages that hold tiles -- 80 instead of 96 at one sample) */
                ex.fail(mcvc_wino_gemm_launch(ga, ws));
                ex.fail(mcvc_wino_dw_launch(ex.wu, grads[c.wi[0]], c.Cout, c.Cin, ws));      // dW += G^T dU G: the chunks add up
            }
            done = true;
        }
    }
    if (!done && c.wino3 && wino_enabled() && ex.wu && (H & 1) == 0 && (W & 1) == 0 && grads[c.wi[0]] && (c.nbr == 1 || grads[c.wi[1]]) &&
        wino43_applies(ex, c, NB, OH, OW)) {
        // ... as F(4x4,3x3)
        static const int en = mcvc_knob("MCVC_WINO3_WGRAD", 1);
        const int K4 = 4 * c.Cin, M = c.cout_tot;
        const int TH = OH / 4, TW = OW / 4;
        const int nbc = wino4_chunk(ex, NB, (long long)TH * TW, M > K4 ? M : K4, 36);
        if (en && nbc && ((long long)nbc * TH * TW >= 32 || ex.force_scheme == 2) && (M % 128) == 0 && (K4 % 64) == 0 && 36LL * M * K4 <= ex.wu_cap) {
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv2;
                xi.N = nb; xi.C = K4; xi.H = OH; xi.W = OW; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 1;
                ex.fail(mcvc_wino43_input_phase_t_launch(xi, H, W, ws));
                WinoXformArgs di{};
                di.x = dy.p + (long long)b0 * dy.sb; di.x_sb = dy.sb; di.x_sc = dy.sc; di.x_sh = dy.sh; di.v = ex.wm2;
                di.N = nb; di.C = M; di.H = OH; di.W = OW; di.TH = TH; di.TW = TW; di.NT = (int)NT; di.NTp = (int)NTp; di.pad = 0;
                ex.fail(mcvc_wino43_dy_t_launch(di, ws));
                WinoGemmArgs ga{};
                ga.a = ex.wm2; ga.a_xi = NTp * M; ga.lda = M;
                ga.b = ex.wv2; ga.b_xi = NTp * K4; ga.ldb = K4;
                ga.c = ex.wu; ga.c_xi = (long long)M * K4; ga.ldc = K4;
                ga.M = M; ga.N = K4; ga.K = (int)((NT + 15) & ~15LL);      /* (rows [NT, NTp) of the tile-major operands are zeros: contract over the 16-row stages that hold tiles -- 80 instead of 96 at one sample) */ ga.nxi = 36;
                ex.fail(mcvc_wino_gemm_launch(ga, ws));
                ex.fail(mcvc_wino43_dw_launch(ex.wu, grads[c.wi[0]], c.nbr == 2 ? grads[c.wi[1]] : nullptr, c.Cout, c.nbr, c.Cin, ws));
            }
            done = true;
        }
    }
    if (!done && c.wino3 && wino_enabled() && ex.wu && (H & 1) == 0 && (W & 1) == 0 && grads[c.wi[0]] && (c.nbr == 1 || grads[c.wi[1]])) {
        // stride-2 5x5 weight gradient in the phase formulation (see conv_fwd): dU[xi][co][4ci+2p+q] over the tiles, both
        // branches (value | gate) in one product, then the 3x3 blocks are scattered back into the two 5x5 gradients
        static const int en = mcvc_knob("MCVC_WINO3_WGRAD", 1);
        const int K4 = 4 * c.Cin, M = c.cout_tot;
        const int TH = (OH + 1) / 2, TW = (OW + 1) / 2;
        const int nbc = wino_chunk(NB, (long long)TH * TW);
        const long long NTc = wino_chunk_tiles(NB, (long long)TH * TW);
        if (en && nbc && (M % 128) == 0 && (K4 % 64) == 0 && 16LL * NTc * M <= ex.wino_cap && 16LL * NTc * K4 <= ex.wino_cap &&
            16LL * M * K4 <= ex.wu_cap) {
            for (int b0 = 0; b0 < NB; b0 += nbc) {
                const int nb = NB - b0 < nbc ? NB - b0 : nbc;
                const long long NT = (long long)nb * TH * TW, NTp = (NT + 31) & ~31LL;
                WinoXformArgs xi{};
                xi.x = x.p + (long long)b0 * x.sb; xi.x_sb = x.sb; xi.x_sc = x.sc; xi.x_sh = x.sh; xi.v = ex.wv2;
                xi.N = nb; xi.C = K4; xi.H = OH; xi.W = OW; xi.TH = TH; xi.TW = TW; xi.NT = (int)NT; xi.NTp = (int)NTp; xi.pad = 1;
                ex.fail(mcvc_wino3_input_phase_t_launch(xi, H, W, ws));
                WinoXformArgs di{};
                di.x = dy.p + (long long)b0 * dy.sb; di.x_sb = dy.sb; di.x_sc = dy.sc; di.x_sh = dy.sh; di.v = ex.wm2;
                di.N = nb; di.C = M; di.H = OH; di.W = OW; di.TH = TH; di.TW = TW; di.NT = (int)NT; di.NTp = (int)NTp; di.pad = 0;
                ex.fail(mcvc_wino3_dy_t_launch(di, ws));
                WinoGemmArgs ga{};
                ga.a = ex.wm2; ga.a_xi = NTp * M; ga.lda = M;
                ga.b = ex.wv2; ga.b_xi = NTp * K4; ga.ldb = K4;
                ga.c = ex.wu; ga.c_xi = (long long)M * K4; ga.ldc = K4;
                ga.M = M; ga.N = K4; ga.K = (int)((NT + 15) & ~15LL);      /* (rows [NT, NTp) of the tile-major operands are zeros: contract over the 16-row stages that hold tiles -- 80 instead of 96 at one sample) */ ga.nxi = 16;
                ex.fail(mcvc_wino_gemm_launch(ga, ws));
                ex.fail(mcvc_wino3_dw_launch(ex.wu, grads[c.wi[0]], c.nbr == 2 ? grads[c.wi[1]] : nullptr, c.Cout, c.nbr, c.Cin, ws));
            }
            done = true;
        }
    }
    if (!done && c.cout_tot == 1 && mcvc_wgrad_cout1_applies(p) && grads[c.wi[0]]) {       // one output channel: VALU kernel
        WgradIO io{x.p, x.sb, x.sc, x.sh, dy.p, dy.sb, dy.sc, dy.sh};
        ex.fail(mcvc_wgrad_cout1_launch(p, NB, io, grads[c.wi[0]], ws));
        done = true;
    }
    for (int br = 0; br < c.nbr && !done; ++br) {
        float* dw = grads[c.wi[br]];
        if (!dw) continue;
        WgradIO io{x.p, x.sb, x.sc, x.sh, dy.p + (long long)br * c.Cout * dy.sc, dy.sb, dy.sc, dy.sh};
        if (mcvc_wgrad_cin1_applies(p) && wgrad_cin1_enabled()) { ex.fail(mcvc_wgrad_cin1_launch(p, NB, io, dw, ws)); continue; }
        // last layer of a backward pass with nothing left for the main stream (conv1 without an input gradient): the gate branch runs
        // there, beside the value branch on the auxiliary stream (atomic accumulation into its own tensor, no slabs to share)
        const bool on_main = br == 1 && ex.br1_on_main && ex.s2 && !mcvc_deterministic() && (long long)c.Cout * c.Cin * c.KH * c.KW <= 65536;
        if (mcvc_wgrad_cin2_applies(p, io)) { ex.fail(mcvc_wgrad_cin2_launch(p, NB, io, dw, on_main ? ex.s : ws)); continue; }
        ex.fail(mcvc_wgrad_launch(p, NB, io, dw, ex.wslabs, ex.wslab_cap, on_main ? ex.s : ws));
    }
    if (ex.s2) {
        hipEvent_t e = pool_event();
        ex.fail(mcvc_event_record(e, ex.s2));
        ex.readers.emplace_back((const void*)dy.p, e);
    }
}

static void conv_bias_grad(Exec& ex, const ConvSpec& c, float* const* grads, int NB, CView dy, int P)
{
    if (ex.dry || !grads || !c.has_bias_grad) return;
    for (int br = 0; br < c.nbr; ++br) {
        float* db = grads[c.bi[br]];
        if (!db) continue;
        ex.fail(mcvc_bias_grad_launch(dy.p + (long long)br * c.Cout * dy.sc, dy.sb, dy.sc, NB, c.Cout, P, db, ex.s));
    }
}

static void pack_spec(Exec& ex, const ConvSpec& c, const float* const* params, float* packed)
{
    const int K = c.Cin * c.KH * c.KW;
    for (int br = 0; br < c.nbr; ++br) {
        const float* w = params[c.wi[br]];
        ex.fail(mcvc_pack_fwd_launch(w, packed + c.off_fwd, c.Cout, K, c.cout_pk, br * c.Cout, ex.s));
        ex.fail(mcvc_copy_launch(params[c.bi[br]], packed + c.off_bias + br * c.Cout, c.Cout, ex.s));
        PackDgradArgs a{};
        a.Cin = c.Cin; a.KH = c.KH; a.KW = c.KW; a.step = c.stride; a.ld = c.merged ? c.mg_ld : c.cin_pk; a.co_off = br * c.Cout; a.ncls = c.ncls;
        a.merged = c.merged; a.mg_kh = c.mg_kh; a.mg_kw = c.mg_kw;
        for (int k = 0; k < c.ncls; ++k) a.cls[k] = c.cls[k];
        ex.fail(mcvc_pack_dgrad_launch(w, packed + c.off_dgrad, a, c.Cout, ex.s));
        if (c.off_tk >= 0) ex.fail(mcvc_pack_trunk_t_launch(w, packed + c.off_tk, c.Cout, c.Cin, c.KW, c.cout_tot * c.KW, br * c.Cout, ex.s));
    }
}

// ---- whole-network pack: job table built once per network kind, resident on the device ---------------------
struct PackTable { std::vector<PackJob> jobs; std::vector<PackDgradArgs> dga; int nblocks = 0; double bytes = 0.0, wbytes = 0.0; };     // wbytes: the written share

static void add_job(PackTable& t, PackJob j, int gx, int gy)
{
    j.block0 = t.nblocks; j.gx = gx;
    t.nblocks += gx * gy;
    t.jobs.push_back(j);
}

// trunk_only: the layer runs on the fused trunk kernels (OIHW forward); only the transposed data-gradient copy is needed
// wino_only: the layer runs on the Winograd kernels in every pass (forward, data-gradient); its direct K-major copies are skipped
// sets: 1 = the copies a FORWARD pass reads (K-major forward copies, biases, forward Winograd sets), 2 = the copies only a BACKWARD pass
// reads (data-gradient copies, transposed trunk copies, data-gradient Winograd sets), 3 = both
// igemm_only: the layer runs on the implicit-GEMM kernels in every pass (the discriminators' stride-2 layers): only the bias, the tap-major
// forward copy and the per-class data-gradient copies are refreshed
static void add_spec_jobs(PackTable& t, const ConvSpec& c, bool trunk_only = false, bool wino_only = false, int sets = 3, bool w4 = true, bool w43 = true,
                          bool igemm_only = false, bool w2 = true)
{
    const int K = c.Cin * c.KH * c.KW;
    const bool fw = (sets & 1) != 0, bw = (sets & 2) != 0;
    for (int br = 0; br < c.nbr; ++br) {
        // (on leaving this iteration: every weight job it added carries its tensor's geometry -- the fused update's tiles, build_upd_table)
        struct Geo { PackTable& t; const ConvSpec& c; size_t first;
                     ~Geo() { for (size_t q = first; q < t.jobs.size(); ++q) if (t.jobs[q].kind != PACK_COPY) {
                                  PackJob& j = t.jobs[q]; j.Cout = c.Cout; j.Cin = c.Cin; j.taps = c.KH * c.KW; } } } geo{t, c, t.jobs.size()};
        const bool skip_direct = (wino_only && (c.wino || c.wino3)) || (igemm_only && c.igemm);
        if (trunk_only && c.off_tk >= 0) {
            if (!bw) continue;
            PackJob q{}; q.kind = PACK_TRUNK_T; q.param = c.wi[br]; q.dst = c.off_tk; q.Cout = c.Cout; q.Cin = c.Cin; q.KW = c.KW;
            q.ld = c.cout_tot * c.KW; q.co_off = br * c.Cout;
            add_job(t, q, cdiv_i(c.Cin, 32), cdiv_i(c.Cout, 32));
            t.bytes += 8.0 * c.Cout * K; t.wbytes += 4.0 * c.Cout * K;
            continue;
        }
        PackJob f{}; f.kind = PACK_FWD; f.param = c.wi[br]; f.dst = c.off_fwd; f.Cout = c.Cout; f.K = K; f.ld = c.cout_pk; f.co_off = br * c.Cout;
        if (!skip_direct && fw) add_job(t, f, cdiv_i(K, 32), cdiv_i(c.Cout, 32));
        PackJob b{}; b.kind = PACK_COPY; b.param = c.bi[br]; b.dst = c.off_bias + br * c.Cout; b.Cout = c.Cout;
        if (fw) add_job(t, b, cdiv_i(c.Cout, 256), 1);
        PackDgradArgs a{};
        a.Cin = c.Cin; a.KH = c.KH; a.KW = c.KW; a.step = c.stride; a.ld = c.merged ? c.mg_ld : c.cin_pk; a.co_off = br * c.Cout; a.ncls = c.ncls;
        a.merged = c.merged; a.mg_kh = c.mg_kh; a.mg_kw = c.mg_kw;
        for (int k = 0; k < c.ncls; ++k) a.cls[k] = c.cls[k];
        PackJob d{}; d.kind = PACK_DGRAD; d.param = c.wi[br]; d.dst = c.off_dgrad; d.dg = (int)t.dga.size();
        if (!skip_direct && bw) {
            t.dga.push_back(a);
            add_job(t, d, cdiv_i(c.Cin, 32), c.Cout);
        }
        if (c.off_dcls >= 0 && bw && !(igemm_only && c.igemm)) {          // exact per-class copies (large-batch data-gradient of the 3x3 stride-2 layers)
            PackDgradArgs u = a;
            u.merged = 0; u.ld = c.cin_pk;
            for (int k = 0; k < c.ncls; ++k) u.cls[k] = c.ucls[k];
            PackJob du{}; du.kind = PACK_DGRAD; du.param = c.wi[br]; du.dst = c.off_dcls; du.dg = (int)t.dga.size();
            t.dga.push_back(u);
            add_job(t, du, cdiv_i(c.Cin, 32), c.Cout);
            t.bytes += 4.0 * 2.0 * c.Cout * K; t.wbytes += 4.0 * c.Cout * K;
        }
        if (c.off_tk >= 0 && bw) {
            PackJob q{}; q.kind = PACK_TRUNK_T; q.param = c.wi[br]; q.dst = c.off_tk; q.Cout = c.Cout; q.Cin = c.Cin; q.KW = c.KW;
            q.ld = c.cout_tot * c.KW; q.co_off = br * c.Cout;
            add_job(t, q, cdiv_i(c.Cin, 32), cdiv_i(c.Cout, 32));
        }
        if (c.igemm && (sets & 4) == 0) {                   // (sets bit 2: skip the implicit-GEMM copies)
            PackJob fi{}; fi.kind = PACK_FWD_TAP; fi.param = c.wi[br]; fi.dst = c.off_ifwd; fi.Cout = c.Cout; fi.K = K; fi.ld = c.cout_pk;
            fi.co_off = br * c.Cout; fi.KW = c.KH * c.KW;
            if (fw) add_job(t, fi, cdiv_i(K, 32), cdiv_i(c.Cout, 32));
            t.bytes += 4.0 * fw * 2.0 * c.Cout * K; t.wbytes += 4.0 * fw * c.Cout * K;
        }
        if (c.wino3) {
            PackJob w3{}; w3.kind = PACK_WINO3_D; w3.param = c.wi[br]; w3.dst = c.off_w3; w3.Cout = c.Cout; w3.Cin = c.Cin; w3.ld = c.mg_ld;
            w3.xi_stride = c.w3_xi; w3.co_off = br * c.Cout;
            if (bw) add_job(t, w3, cdiv_i(c.Cin, 256), c.Cout);
            PackJob w3f{}; w3f.kind = PACK_WINO3_F; w3f.param = c.wi[br]; w3f.dst = c.off_w3f; w3f.Cout = c.Cout; w3f.Cin = c.Cin; w3f.ld = c.cout_pk;
            w3f.xi_stride = c.w3f_xi; w3f.co_off = br * c.Cout;
            if (fw) add_job(t, w3f, cdiv_i(c.Cout, 256), c.Cin);
            t.bytes += 4.0 * (fw + bw) * (25.0 + 64.0) * c.Cout * c.Cin; t.wbytes += 4.0 * (fw + bw) * 64.0 * c.Cout * c.Cin;
        }
        if (c.wino3 && w43) {
            PackJob w3{}; w3.kind = PACK_WINO43_D; w3.param = c.wi[br]; w3.dst = c.off_w43; w3.Cout = c.Cout; w3.Cin = c.Cin; w3.ld = c.mg_ld;
            w3.xi_stride = c.w3_xi; w3.co_off = br * c.Cout;
            if (bw) add_job(t, w3, cdiv_i(c.Cin, 256), c.Cout);
            PackJob w3f{}; w3f.kind = PACK_WINO43_F; w3f.param = c.wi[br]; w3f.dst = c.off_w43f; w3f.Cout = c.Cout; w3f.Cin = c.Cin; w3f.ld = c.cout_pk;
            w3f.xi_stride = c.w3f_xi; w3f.co_off = br * c.Cout;
            if (fw) add_job(t, w3f, cdiv_i(c.Cout, 256), c.Cin);
            t.bytes += 4.0 * (fw + bw) * (25.0 + 144.0) * c.Cout * c.Cin; t.wbytes += 4.0 * (fw + bw) * 144.0 * c.Cout * c.Cin;
        }
        if (c.wino && w4) {
            PackJob wf{}; wf.kind = PACK_WINO4_F; wf.param = c.wi[br]; wf.dst = c.off_w4f; wf.Cout = c.Cout; wf.Cin = c.Cin; wf.ld = c.cout_pk;
            wf.xi_stride = c.wf_xi; wf.co_off = br * c.Cout;
            if (fw) add_job(t, wf, cdiv_i(c.Cout, 256), c.Cin);
            PackJob wd{}; wd.kind = PACK_WINO4_D; wd.param = c.wi[br]; wd.dst = c.off_w4d; wd.Cout = c.Cout; wd.Cin = c.Cin; wd.ld = c.cin_pk;
            wd.xi_stride = c.wd_xi; wd.co_off = br * c.Cout;
            if (bw) add_job(t, wd, cdiv_i(c.Cin, 256), c.Cout);
            t.bytes += 4.0 * (fw + bw) * (25.0 + 64.0) * c.Cout * c.Cin; t.wbytes += 4.0 * (fw + bw) * 64.0 * c.Cout * c.Cin;
        }
        if (c.wino && w2) {
            PackJob wf{}; wf.kind = PACK_WINO_F; wf.param = c.wi[br]; wf.dst = c.off_wf; wf.Cout = c.Cout; wf.Cin = c.Cin; wf.ld = c.cout_pk;
            wf.xi_stride = c.wf_xi; wf.co_off = br * c.Cout;
            if (fw) add_job(t, wf, cdiv_i(c.Cout, 256), c.Cin);
            PackJob wd{}; wd.kind = PACK_WINO_D; wd.param = c.wi[br]; wd.dst = c.off_wd; wd.Cout = c.Cout; wd.Cin = c.Cin; wd.ld = c.cin_pk;
            wd.xi_stride = c.wd_xi; wd.co_off = br * c.Cout;
            if (bw) add_job(t, wd, cdiv_i(c.Cin, 256), c.Cout);
            t.bytes += 4.0 * (fw + bw) * (25.0 + 36.0) * c.Cout * c.Cin; t.wbytes += 4.0 * (fw + bw) * 36.0 * c.Cout * c.Cin;
        }
        t.bytes += skip_direct ? 8.0 * c.Cout : 4.0 * ((c.off_tk >= 0 ? 6.0 : 4.0) * c.Cout * K + 2.0 * c.Cout);
        t.wbytes += skip_direct ? 4.0 * c.Cout : 4.0 * ((c.off_tk >= 0 ? 3.0 : 2.0) * c.Cout * K + c.Cout);
    }
}

struct DevPackTable { PackJob* jobs; PackDgradArgs* dga; int njobs, nblocks; double bytes; std::vector<int> used; };

// one upload per (device, network kind); the first call must happen outside stream capture (it allocates)
template <class Build>
static const DevPackTable* dev_pack_table(int kind, Build&& build, int* err)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, DevPackTable> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { *err = MCVC_ERR_INVALID; return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, kind});
    if (it != cache.end()) return &it->second;
    PackTable t;
    build(t);
    DevPackTable d{};
    d.njobs = (int)t.jobs.size(); d.nblocks = t.nblocks; d.bytes = t.bytes;
    hipError_t e = hipMalloc((void**)&d.jobs, t.jobs.size() * sizeof(PackJob));
    if (e == hipSuccess) e = hipMalloc((void**)&d.dga, t.dga.size() * sizeof(PackDgradArgs));
    if (e == hipSuccess) e = hipMemcpy(d.jobs, t.jobs.data(), t.jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.dga, t.dga.data(), t.dga.size() * sizeof(PackDgradArgs), hipMemcpyHostToDevice);
    if (e != hipSuccess) { *err = (int)e; return nullptr; }
    for (const PackJob& j : t.jobs) d.used.push_back(j.param);
    return &cache.emplace(std::make_pair(dev, kind), std::move(d)).first->second;
}

static int pack_net(const DevPackTable* t, const float* const* params, float* packed, hipStream_t s)
{
    PackPtrs ptrs{};
    for (int i : t->used) { if (i < 0 || i >= 128) return MCVC_ERR_INVALID; ptrs.p[i] = params[i]; }
    return mcvc_pack_net_launch(t->jobs, t->njobs, t->nblocks, t->dga, ptrs, packed, t->bytes, s);
}

// ---- optimizer step fused with the re-pack (pack.h: UpdOwner) -----------------------------------------------------------------------
// The job table of a re-pack (same `build` as dev_pack_table) is re-grouped by parameter: every tensor of the parameter range gets ONE owner
// -- a tiled owner whose workgroups update a (CB x IB x taps) block of filters and emit all packed forms of that block, or a flat owner
// (biases, norm affine parameters, tensors nothing is packed from) -- so that one launch is optimizer.step() AND the refresh of every
// copy the kernels read (reference train.py:242,299).
struct DevUpdTable { UpdOwner* owners; PackJob* jobs; PackDgradArgs* dga; int nown, nblocks; double adam_floats, pack_bytes; std::vector<int> used;
                     std::vector<long long> sizes; };      // sizes[k]: element count of parameter used[k] the tiles were built for

template <class Build, class InRange>
static const DevUpdTable* dev_upd_table(int kind, Build&& build, const long long* numel, int nparams, InRange&& in_range, int* err)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, DevUpdTable> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { *err = MCVC_ERR_INVALID; return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, kind});
    if (it != cache.end()) return &it->second;
    PackTable t;
    build(t);
    std::vector<PackJob> jobs;                  // re-ordered: the emits of one parameter are consecutive
    std::vector<UpdOwner> owners;
    DevUpdTable d{};
    int nblocks = 0;
    for (int prm = 0; prm < nparams && prm < 128; ++prm) {
        std::vector<PackJob> mine;
        for (const PackJob& j : t.jobs) if (j.param == prm) mine.push_back(j);
        if (!in_range(prm)) { if (!mine.empty()) { *err = MCVC_ERR_INVALID; return nullptr; } continue; }
        if (numel[prm] <= 0) { if (!mine.empty()) { *err = MCVC_ERR_INVALID; return nullptr; } continue; }       // (dead parameter)
        UpdOwner o{};
        o.param = prm; o.block0 = nblocks; o.e0 = (int)jobs.size(); o.ne = (int)mine.size();
        bool weight = false;
        for (const PackJob& j : mine) weight = weight || j.kind != PACK_COPY;
        if (!weight) {
            o.flat = 1; o.Cout = (int)numel[prm]; o.Cin = o.taps = 1; o.gx = 1;
            for (const PackJob& j : mine) if (j.Cout != o.Cout) { *err = MCVC_ERR_INVALID; return nullptr; }
            nblocks += cdiv_i(o.Cout, 256);
        } else {
            const PackJob& j0 = mine[0];
            o.Cout = j0.Cout; o.Cin = j0.Cin; o.taps = j0.taps;
            for (const PackJob& j : mine) if (j.kind == PACK_COPY || j.Cout != o.Cout || j.Cin != o.Cin || j.taps != o.taps) { *err = MCVC_ERR_INVALID; return nullptr; }
            // (the Winograd weight transforms read ONE 5 x 5 filter = 25 consecutive LDS words: upd_emit_filters)
            for (const PackJob& j : mine) if (j.kind >= PACK_WINO_F && j.kind <= PACK_WINO43_F && j.taps != 25) { *err = MCVC_ERR_INVALID; return nullptr; }
            if ((long long)o.Cout * o.Cin * o.taps != numel[prm]) { *err = MCVC_ERR_INVALID; return nullptr; }
            // tile: 32 x 16 filters of 25 taps (the Winograd layers: two filters per thread), 32 x 32 of 9, 32 x 64 of 3, 32 x 128 of 1;
            // the 5 x 15 edge layers (2 input channels / 1 output channel) as they come
            int cb = 32, ib = 32;
            // (25 taps: 32 x 16 filters = two per thread, 52 KB of LDS -- full 128-byte runs in the forward-type copies; measured against
            //  16 x 16: the family 0.733 -> 0.716 ms per bs=1 iteration, 1.26 -> 1.19 at bs=8.  MCVC_UPD_CB25=16 for the A/B)
            static const int cb25 = mcvc_knob("MCVC_UPD_CB25", 32);
            if (o.taps == 25) { cb = cb25 == 16 ? 16 : 32; ib = 16; }
            else if (o.taps == 3) ib = 64;
            else if (o.taps == 1) ib = 128;
            else if (o.taps > 25) ib = 16;
            o.CB = cb < o.Cout ? cb : o.Cout; o.IB = ib < o.Cin ? ib : o.Cin;
            if ((size_t)o.CB * mcvc_upd_pitch(o.IB, o.taps) * sizeof(float) > kUpdLds) { *err = MCVC_ERR_INVALID; return nullptr; }
            o.vec4 = ((o.IB * o.taps) % 4 == 0 && (o.Cin % o.IB) == 0 && ((long long)o.Cin * o.taps) % 4 == 0) ? 1 : 0;
            o.gx = cdiv_i(o.Cin, o.IB);
            nblocks += o.gx * cdiv_i(o.Cout, o.CB);
        }
        d.adam_floats += (double)numel[prm];
        for (const PackJob& j : mine) jobs.push_back(j);
        owners.push_back(o);
        d.used.push_back(prm); d.sizes.push_back(numel[prm]);
    }
    if (jobs.size() != t.jobs.size() || owners.empty()) { *err = MCVC_ERR_INVALID; return nullptr; }
    d.nown = (int)owners.size(); d.nblocks = nblocks;
    d.pack_bytes = t.wbytes;
    if (jobs.empty()) jobs.push_back(PackJob{});
    if (t.dga.empty()) t.dga.push_back(PackDgradArgs{});
    hipError_t e = hipMalloc((void**)&d.owners, owners.size() * sizeof(UpdOwner));
    if (e == hipSuccess) e = hipMalloc((void**)&d.jobs, jobs.size() * sizeof(PackJob));
    if (e == hipSuccess) e = hipMalloc((void**)&d.dga, t.dga.size() * sizeof(PackDgradArgs));
    if (e == hipSuccess) e = hipMemcpy(d.owners, owners.data(), owners.size() * sizeof(UpdOwner), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.dga, t.dga.data(), t.dga.size() * sizeof(PackDgradArgs), hipMemcpyHostToDevice);
    if (e != hipSuccess) { *err = (int)e; return nullptr; }
    return &cache.emplace(std::make_pair(dev, kind), std::move(d)).first->second;
}

// the optimizer arguments of the fused update: the gradient(s) and the moments live at the same offsets of their flat buffers as the
// parameters do in theirs (engine.py _FlatGroup), so one float offset per buffer locates them from a parameter pointer
struct UpdOpt { const float* flat; const float* grad; const float* grad2; const float* exp_avg; const float* exp_avg_sq;
                float lr, beta1, beta2, eps; int step; float grad_scale; int zero_grads; };

static int update_net(const DevUpdTable* t, const float* const* params, const long long* numel, float* packed, const UpdOpt& o, hipStream_t s)
{
    for (size_t k = 0; k < t->used.size(); ++k) if (numel[t->used[k]] != t->sizes[k]) return MCVC_ERR_INVALID;      // (the table is cached per configuration)
    if (!o.flat || !o.grad || !o.exp_avg || !o.exp_avg_sq || o.step < 1) return MCVC_ERR_INVALID;
    if ((((uintptr_t)o.flat | (uintptr_t)o.grad | (uintptr_t)o.grad2 | (uintptr_t)o.exp_avg | (uintptr_t)o.exp_avg_sq) & 15) != 0) return MCVC_ERR_INVALID;
    PackPtrs ptrs{};
    for (int i : t->used) { if (i < 0 || i >= 128 || !params[i] || (((uintptr_t)params[i]) & 15) != 0) return MCVC_ERR_INVALID; ptrs.p[i] = params[i]; }
    UpdAdam ad{};
    ad.d_g = o.grad - o.flat; ad.d_m = o.exp_avg - o.flat; ad.d_v = o.exp_avg_sq - o.flat;
    ad.has_g2 = o.grad2 ? 1 : 0; ad.d_g2 = o.grad2 ? o.grad2 - o.flat : 0;
    ad.zero = o.zero_grads;
    ad.c = mcvc_adam_coef(o.lr, o.beta1, o.beta2, o.eps, o.step, o.grad_scale);
    const double bytes = (28.0 + (o.grad2 ? 4.0 : 0.0) + (o.zero_grads ? (o.grad2 ? 8.0 : 4.0) : 0.0)) * t->adam_floats + t->pack_bytes;
    return mcvc_update_net_launch(t->owners, t->nown, t->nblocks, t->jobs, t->dga, ptrs, packed, ad, bytes, s);
}

// ---- fused small-batch trunk layer (trunk_kernels.hip) ------------------------------------------------
static bool trunk_enabled()
{
    static const int en = mcvc_knob("MCVC_TRUNK", 1);
    return en != 0;
}

// persistent 13-layer forward (trunk_fwd_net_kernel) instead of one fused launch per layer; knob MCVC_TRUNK_NET=0 for A/B runs
static int g_trunk_net = mcvc_knob("MCVC_TRUNK_NET", 1) != 0;
static bool trunk_net_enabled() { return g_trunk_net != 0; }
static bool trunk_bwd_net_enabled() { return g_trunk_net == 1; }          // (2 = forward only)

// conv1d + bias + IN (+GLU | +residual); input / conv_out in trunk layout [C][B][W4]; y plane (b, c) at y + b*y_sn + c*y_sc
static bool trunk_fwd(Exec& ex, const ConvSpec& c, const float* const* P, int g0, int be0, int g1, int be1, const float* x, float* conv_out,
                      float* stats, float* y, long long y_sn, long long y_sc, const float* res, int B, int W4)
{
    const int mode = (c.nbr == 2) ? TRUNK_IN_GLU : TRUNK_IN;
    if (!trunk_enabled() || c.KH != 1 || !mcvc_trunk_applies(c.Cin, c.KW, c.Cout, B, W4, mode, 1)) return false;
    if (ex.dry) return true;
    TrunkArgs a{};
    a.a0 = P[c.wi[0]]; a.bias0 = P[c.bi[0]]; a.gamma0 = P[g0]; a.beta0 = P[be0];
    if (c.nbr == 2) { a.a1 = P[c.wi[1]]; a.bias1 = P[c.bi[1]]; a.gamma1 = P[g1]; a.beta1 = P[be1]; }
    a.x = x; a.x_sc = (long long)B * W4; a.x_sb = W4;
    a.Cin = c.Cin; a.KW = c.KW; a.K = c.Cin * c.KW; a.M = c.Cout; a.Mtot = c.cout_tot; a.B = B; a.T4 = W4; a.N = B * W4;
    a.conv_out = conv_out; a.c_sc = (long long)B * W4; a.c_sb = W4;
    a.stats = stats; a.y = y; a.res = res; a.y_sn = y_sn; a.y_sc = y_sc; a.eps = kInEps; a.mode = mode;
    ex.fail(mcvc_trunk_launch(a, 1, ex.s));
    return true;
}

// largest K split (power-of-two multiples of `start`) the kernel accepts that still keeps the grid below ~512 workgroups
static int trunk_pick_ksplit(int Cin, int KW, int M, int B, int W4, int must_split)
{
    int best = 0;
    for (int ks = 1; ks <= 64; ++ks) {
        if (!mcvc_trunk_applies(Cin, KW, M, B, W4, TRUNK_PLAIN, ks)) continue;
        if (best == 0) best = ks;
        if ((M / 16) * ks <= 512) best = ks;
    }
    (void)must_split;
    return best;
}

// dX[ci][b][t] (+)= sum_{co,kw} W[co][ci][kw] dY[co][b][t + pw - kw]  through the transposed pack.  Large K (or an
// accumulating destination) is split over workgroups that add atomically; a non-accumulating destination is zeroed first.
struct TrunkPre {              // fused InstanceNorm backward in front of the data-gradient (trunk.h)
    int kind;                  // 1 plain IN, 2 IN + gated GLU
    const float* x; const float* stats;
    const float* g0; const float* b0; const float* g1; const float* b1;
    float* out;                // X' = gradient w.r.t. the conv output (the weight gradient reads it)
    float* dg0; float* db0; float* dg1; float* db1;
    int xB;                    // samples per channel of x (0 = the pass's batch; larger: backward over a prefix of the forward pass's samples)
};

static bool trunk_dgrad(Exec& ex, const ConvSpec& c, const float* packed, const float* dy, float* dx, int accumulate, int B, int W4,
                        int* nsplit = nullptr, const TrunkPre* pre = nullptr)
{
    if (!trunk_enabled() || c.off_tk < 0) return false;
    int ks = 1;
    if (!accumulate && mcvc_trunk_applies(c.cout_tot, c.KW, c.Cin, B, W4, TRUNK_PLAIN, 1)) ks = 1;
    else ks = trunk_pick_ksplit(c.cout_tot, c.KW, c.Cin, B, W4, 0);
    if (ks < 1) return false;
    if (!accumulate && ks > 1 && !nsplit) return false;             // slabs need a consumer that sums them
    const long long tot = (long long)c.Cin * B * W4;
    // deterministic mode: an accumulating K split leaves dx alone and writes ks private slabs; the consumer sums dx + slabs
    const bool slab_all = accumulate && ks > 1 && nsplit && mcvc_deterministic();
    if (ks > 1 && (!accumulate || nsplit)) {
        const long long need = (long long)(accumulate ? ks : ks - 1) * tot;      // (sized for both modes)
        if (need > ex.slab_need) ex.slab_need = need;
        if (!ex.dry && need > ex.slab_cap) { ex.fail(MCVC_ERR_WORKSPACE); return true; }
    }
    if (nsplit) *nsplit = slab_all ? ks + 1 : (accumulate ? 1 : ks);
    if (ex.dry) return true;
    TrunkArgs a{};
    a.a0 = packed + c.off_tk;
    a.x = dy; a.x_sc = (long long)B * W4; a.x_sb = W4;
    a.Cin = c.cout_tot; a.KW = c.KW; a.K = c.cout_tot * c.KW; a.M = c.Cin; a.Mtot = c.Cin; a.B = B; a.T4 = W4; a.N = B * W4;
    a.conv_out = dx; a.c_sc = (long long)B * W4; a.c_sb = W4; a.accumulate = accumulate; a.mode = TRUNK_PLAIN;
    a.slabs = ex.slabs; a.slab_stride = tot; a.slab_all = slab_all ? 1 : 0;
    if (pre) {
        a.pre = pre->kind; a.pre_C = (pre->kind == 2) ? c.cout_tot / 2 : c.cout_tot;
        a.pre_x = pre->x; a.pre_xB = pre->xB; a.pre_stats = pre->stats; a.pre_gamma0 = pre->g0; a.pre_beta0 = pre->b0; a.pre_gamma1 = pre->g1; a.pre_beta1 = pre->b1;
        a.pre_out = pre->out; a.pre_dgamma0 = pre->dg0; a.pre_dbeta0 = pre->db0; a.pre_dgamma1 = pre->dg1; a.pre_dbeta1 = pre->db1;
        wait_readers(ex, pre->out);
    }
    ex.fail(mcvc_trunk_launch(a, ks, ex.s));
    return true;
}

// K too large for one workgroup (conv2dto1d, K = 5120): K-split workgroups write private slabs (split 0 adds the bias) that
// the InstanceNorm launch sums -- no atomics, no zero-fill.  Weights are read straight from the OIHW parameter.
static bool trunk_fwd_ksplit(Exec& ex, const ConvSpec& c, const float* const* P, const float* x, float* conv_out, int B, int W4, int* nsplit)
{
    if (!trunk_enabled() || c.KH != 1 || c.nbr != 1) return false;
    const int ks = trunk_pick_ksplit(c.Cin, c.KW, c.Cout, B, W4, 1);
    if (ks < 1) return false;
    const long long tot = (long long)c.Cout * B * W4;
    const long long need = (long long)(ks - 1) * tot;
    if (need > ex.slab_need) ex.slab_need = need;
    if (!ex.dry && need > ex.slab_cap) { ex.fail(MCVC_ERR_WORKSPACE); return true; }
    *nsplit = ks;
    if (ex.dry) return true;
    TrunkArgs a{};
    a.a0 = P[c.wi[0]]; a.bias0 = P[c.bi[0]];
    a.x = x; a.x_sc = (long long)B * W4; a.x_sb = W4;
    a.Cin = c.Cin; a.KW = c.KW; a.K = c.Cin * c.KW; a.M = c.Cout; a.Mtot = c.Cout; a.B = B; a.T4 = W4; a.N = B * W4;
    a.conv_out = conv_out; a.c_sc = (long long)B * W4; a.c_sb = W4; a.accumulate = 0; a.mode = TRUNK_PLAIN;
    a.slabs = ex.slabs; a.slab_stride = tot;
    ex.fail(mcvc_trunk_launch(a, ks, ex.s));
    return true;
}

// ---- norm wrappers -------------------------------------------------------------------------------------
struct NormP { const float* g[2]; const float* b[2]; float* dg[2]; float* db[2]; };

static NormP normp(const float* const* params, float* const* grads, int g0, int b0, int g1 = -1, int b1 = -1)
{
    NormP n{};
    if (!params) return n;                       // dry (sizing) run
    n.g[0] = params[g0]; n.b[0] = params[b0];
    if (g1 >= 0) { n.g[1] = params[g1]; n.b[1] = params[b1]; }
    if (grads) {
        n.dg[0] = grads[g0]; n.db[0] = grads[b0];
        if (g1 >= 0) { n.dg[1] = grads[g1]; n.db[1] = grads[b1]; }
    }
    return n;
}

static int fuse_wino_norm()
{
    static const int en = mcvc_knob("MCVC_FUSE_WINO_NORM", 1);
    return en;
}

static void norm_fwd(Exec& ex, float* x, long long x_sn, long long x_sc, long long x_total, int nslab, const NormP& np, float* stats,
                     float* y, long long y_sn, long long y_sc, int y_sh, const float* res, int N, int C, int H, int W, int act)
{
    if (ex.dry) { ex.y_xs_pw = 0; return; }
    NormArgs a{};
    a.x = x; a.x_slabs = ex.slabs; a.x_sn = x_sn; a.x_sc = x_sc; a.slab_stride = x_total; a.nslab = nslab;
    a.gamma[0] = np.g[0]; a.gamma[1] = np.g[1]; a.beta[0] = np.b[0]; a.beta[1] = np.b[1];
    a.stats = stats; a.y = y; a.res = res; a.y_sn = y_sn; a.y_sc = y_sc; a.y_sh = y_sh;
    a.N = N; a.C = C; a.H = H; a.W = W; a.act = act; a.eps = kInEps;
    if (ex.y_xs_pw) { a.y_xs = 1; a.xs_pw = ex.y_xs_pw; a.xs_plane = ex.y_xs_plane; ex.y_xs_pw = 0; }
    if (ex.pend_pts) {
        const int pts = ex.pend_pts;
        ex.pend_pts = 0;
        if (mcvc_norm_fwd_wino_applies(a, ex.pend, pts)) { ex.fail(mcvc_norm_fwd_wino_launch(a, ex.pend, pts, ex.s)); return; }
        ex.fail(pts == 16 ? mcvc_wino3_output_launch(ex.pend, ex.s) : pts == 36 ? mcvc_wino_output_launch(ex.pend, ex.s)
                : pts == 43 ? mcvc_wino43_output_launch(ex.pend, ex.s) : mcvc_wino4_output_launch(ex.pend, ex.s));
    }
    ex.fail(mcvc_norm_fwd_launch(a, ex.s));
}

static void norm_bwd(Exec& ex, const float* x, long long x_sn, long long x_sc, const NormP& np, const float* stats,
                     float* dy, long long y_sn, long long y_sc, int y_sh, long long dy_total, int nslab,
                     float* dx, long long dx_sn, long long dx_sc, int dx_sh, int unshuffle, int N, int C, int H, int W, int act)
{
    if (ex.dry) { ex.dx_pitch = 0; return; }
    wait_readers(ex, dx);
    NormBwdArgs a{};
    a.x = x; a.x_sn = x_sn; a.x_sc = x_sc;
    a.gamma[0] = np.g[0]; a.gamma[1] = np.g[1]; a.beta[0] = np.b[0]; a.beta[1] = np.b[1];
    a.stats = stats; a.dy = dy; a.dy_slabs = ex.slabs; a.y_sn = y_sn; a.y_sc = y_sc; a.y_sh = y_sh; a.slab_stride = dy_total; a.nslab = nslab;
    a.dx = dx; a.dx_sn = dx_sn; a.dx_sc = dx_sc; a.dx_sh = dx_sh; a.unshuffle = unshuffle;
    a.dgamma[0] = np.dg[0]; a.dgamma[1] = np.dg[1]; a.dbeta[0] = np.db[0]; a.dbeta[1] = np.db[1];
    a.N = N; a.C = C; a.H = H; a.W = W; a.act = act;
    a.dx_pitch = ex.dx_pitch; ex.dx_pitch = 0;
    ex.fail(mcvc_norm_bwd_launch(a, ex.s));
}

static void act_fwd(Exec& ex, float* x, long long x_total, int nslab, float* y, int N, int C, int P, int act)
{
    if (ex.dry) return;
    ActArgs a{}; a.x = x; a.x_slabs = ex.slabs; a.slab_stride = x_total; a.nslab = nslab; a.y = y; a.N = N; a.C = C; a.P = P; a.act = act;
    ex.fail(mcvc_act_fwd_launch(a, ex.s));
}

static void act_bwd(Exec& ex, const float* x, float* dy, long long dy_total, int nslab, float* dx, int N, int C, int P, int act)
{
    if (ex.dry) return;
    wait_readers(ex, dx);
    ActBwdArgs a{}; a.x = x; a.dy = dy; a.dy_slabs = ex.slabs; a.slab_stride = dy_total; a.nslab = nslab; a.dx = dx; a.N = N; a.C = C; a.P = P; a.act = act;
    ex.fail(mcvc_act_bwd_launch(a, ex.s));
}

// =================================================================================================
// Generator
// =================================================================================================
struct GenNet {
    ConvSpec conv1, ds1, ds2, c2d1d, res_vg[6], res_out[6], c1d2d, up1, up2, last;
    long long packed_floats;
};

static GenNet build_gen()
{
    GenNet g{};
    long long cur = 0;
    g.conv1 = mk(2, 128, 2, 5, 15, 1, 2, 7, 0, 1, 2, 3, 1);            // model.py:116-126 (no norm -> bias grads live)
    g.ds1 = mk(128, 256, 2, 5, 5, 2, 2, 2, 4, 5, 8, 9, 0);              // :129-133
    g.ds2 = mk(256, 256, 2, 5, 5, 2, 2, 2, 12, 13, 16, 17, 0);          // :135-139
    g.c2d1d = mk(5120, 256, 1, 1, 1, 1, 0, 0, 20, 21, -1, -1, 0);       // :142-146
    for (int i = 0; i < 6; ++i) {                                       // :151-180
        const int b = 24 + 12 * i;
        g.res_vg[i] = mk(256, 512, 2, 1, 3, 1, 0, 1, b + 0, b + 1, b + 4, b + 5, 0);
        g.res_out[i] = mk(512, 256, 1, 1, 3, 1, 0, 1, b + 8, b + 9, -1, -1, 0);
    }
    g.c1d2d = mk(256, 5120, 1, 1, 1, 1, 0, 0, 96, 97, -1, -1, 0);       // :183-187
    g.up1 = mk(256, 1024, 1, 5, 5, 1, 2, 2, 104, 105, -1, -1, 1);       // :192-196 (PixelShuffle before the norm -> bias grad live)
    g.up2 = mk(256, 512, 1, 5, 5, 1, 2, 2, 100, 101, -1, -1, 1);        // :200-204 (named convLayer.* in named_parameters)
    g.last = mk(128, 1, 1, 5, 15, 1, 2, 7, 108, 109, -1, -1, 1);        // :207-211
    ConvSpec* all[] = {&g.conv1, &g.ds1, &g.ds2, &g.c2d1d, &g.c1d2d, &g.up1, &g.up2, &g.last};
    for (ConvSpec* c : all) spec_finalize(*c, cur);
    for (int i = 0; i < 6; ++i) { spec_finalize(g.res_vg[i], cur); spec_finalize(g.res_out[i], cur); }
    g.packed_floats = cur;
    return g;
}
static const GenNet& gen_net() { static const GenNet g = build_gen(); return g; }

struct GenDims {
    int B, T, W2, W4, Wu1, Wu2;
    long long big;       // largest activation / conv-output tensor (floats)
};
static GenDims gen_dims(int B, int T)
{
    GenDims d{};
    d.B = B; d.T = T;
    d.W2 = conv_out(T, 5, 2, 2);
    d.W4 = conv_out(d.W2, 5, 2, 2);
    d.Wu1 = 2 * d.W4; d.Wu2 = 4 * d.W4;
    const int wmax = d.T > d.Wu2 ? d.T : d.Wu2;
    d.big = (long long)B * 256 * 80 * wmax;
    return d;
}

struct GenStash {
    long long xin, c1, y1, c2, s2, y2, c3, s3, y3, c4, s4, y4;
    struct { long long ca, sa, ya, cb, sb, y; } r[6];
    long long c6, s6, y6, c7, s7, y7, c8, s8, y8, total;
};
static GenStash gen_stash(const GenDims& d)
{
    GenStash s{};
    long long cur = 0;
    auto take = [&](long long n) { const long long o = cur; cur += (n + 3) & ~3LL; return o; };
    const long long B = d.B;
    s.xin = take(B * 2 * 80 * d.T);
    s.c1 = take(B * 256 * 80 * d.T);   s.y1 = take(B * 128 * 80 * d.T);
    s.c2 = take(B * 512 * 40 * d.W2);  s.s2 = take(B * 512 * 2);  s.y2 = take(B * 256 * 40 * d.W2);
    s.c3 = take(B * 512 * 20 * d.W4);  s.s3 = take(B * 512 * 2);  s.y3 = take(B * 5120 * d.W4);
    s.c4 = take(B * 256 * d.W4);       s.s4 = take(B * 256 * 2);  s.y4 = take(B * 256 * d.W4);
    for (int i = 0; i < 6; ++i) {
        s.r[i].ca = take(B * 1024 * d.W4); s.r[i].sa = take(B * 1024 * 2); s.r[i].ya = take(B * 512 * d.W4);
        s.r[i].cb = take(B * 256 * d.W4);  s.r[i].sb = take(B * 256 * 2);  s.r[i].y = take(B * 256 * d.W4);
    }
    s.c6 = take(B * 5120 * d.W4);      s.s6 = take(B * 5120 * 2); s.y6 = take(B * 5120 * d.W4);
    s.c7 = take(B * 256 * 40 * d.Wu1); s.s7 = take(B * 256 * 2);  s.y7 = take(B * 256 * 40 * d.Wu1);
    s.c8 = take(B * 128 * 80 * d.Wu2); s.s8 = take(B * 128 * 2);  s.y8 = take(B * 128 * 80 * d.Wu2);
    s.total = cur;
    return s;
}

struct GenScratch { long long ga, gb, gb2, dh, dt1, dt1b, dt2, dt3, dt3b, dtx1, dtx3, wv, wm, wino_floats, wv2, wm2, wu, wu_floats, sync, slabs; };
static GenScratch gen_scratch(const GenDims& d)
{
    GenScratch s{};
    long long cur = 0;
    auto take = [&](long long n) { const long long o = cur; cur += (n + 3) & ~3LL; return o; };
    // first, so that its place does not depend on the batch size: callers share one scratch buffer between passes of different batch
    // (the trainer's B and 2B passes), and the sticky error word (mcvc_gen_trunk_fault) must not land inside another layout's data
    s.sync = take(2 * MCVC_TRUNK_SYNC_WORDS);            // forward | backward persistent trunk kernels
    s.ga = take(d.big); s.gb = take(d.big); s.gb2 = take(d.big);
    s.dh = take((long long)256 * d.B * d.W4); s.dt1 = take((long long)1024 * d.B * d.W4); s.dt1b = take((long long)1024 * d.B * d.W4);
    s.dt2 = take((long long)512 * d.B * d.W4); s.dt3 = take((long long)256 * d.B * d.W4); s.dt3b = take((long long)256 * d.B * d.W4);
    // small batch: one conv-output gradient buffer per residual layer, so that ONE batched launch computes all their weight gradients
    s.dtx1 = take(mcvc_wgrad_smallk_batch_applies(d.B, d.W4) ? 6LL * 1024 * d.B * d.W4 : 0);
    s.dtx3 = take(mcvc_wgrad_smallk_batch_applies(d.B, d.W4) ? 6LL * 256 * d.B * d.W4 : 0);
    // Winograd V / M of upSample1 (1024 ch, tiles of a 20 x W4 image) and upSample2 (512 ch, 40 x 2*W4)
    {
        // (a batch beyond kWinoMaxTiles tiles runs in chunks of samples: the workspaces are sized for one chunk)
        const long long nt1 = wino_chunk_tiles(d.B, 10LL * ((d.W4 + 1) / 2)), nt2 = wino_chunk_tiles(d.B, 20LL * ((d.Wu1 + 1) / 2));
        const long long a1 = 36LL * 1024 * nt1, a2 = 36LL * 512 * nt2;
        s.wino_floats = wino_enabled() ? (a1 > a2 ? a1 : a2) : 0;
        s.wv = take(s.wino_floats); s.wm = take(s.wino_floats);
        s.wv2 = take(s.wino_floats); s.wm2 = take(s.wino_floats);
        s.wu_floats = wino_enabled() ? 36LL * 1024 * 1024 : 0;        // dU of downSample2 in phase form, 36 points of F(4x4,3x3) (> upSample1's 64 x 1024 x 256)
        s.wu = take(s.wu_floats);
    }
    s.slabs = cur;
    return s;
}

static void gen_forward_impl(Exec& ex, const float* const* P, const float* packed, const float* x, const float* mask, float* out,
                             float* st, const GenDims& d)
{
    ex.params = P;
    const GenNet& g = gen_net();
    const GenStash o = gen_stash(d);
    const int B = d.B, T = d.T, W2 = d.W2, W4 = d.W4, Wu1 = d.Wu1, Wu2 = d.Wu2;
    const long long BT4 = (long long)B * W4;
    int ns = 1;
    // ---- model.py:241-242  stack(x*mask, mask) -> gated 5x15 conv (value|gate in one launch)
    if (!ex.dry) ex.fail(mcvc_prep_input_launch(x, mask, st + o.xin, B, 80 * T, ex.s));
    conv_fwd(ex, g.conv1, packed, B, 80, T, CView{st + o.xin, 2LL * 80 * T, 80LL * T, T}, View{st + o.c1, 256LL * 80 * T, 80LL * T, T},
             (long long)B * 256 * 80 * T, 0, 1, &ns);
    act_fwd(ex, st + o.c1, (long long)B * 256 * 80 * T, ns, st + o.y1, B, 128, 80 * T, ACT_GLU);
    // ---- :245  downSample1 (5x5 s2, IN, GLU)
    ex.fuse_next = fuse_wino_norm();
    conv_fwd(ex, g.ds1, packed, B, 80, T, CView{st + o.y1, 128LL * 80 * T, 80LL * T, T}, View{st + o.c2, 512LL * 40 * W2, 40LL * W2, W2},
             (long long)B * 512 * 40 * W2, 0, 1, &ns);
    ex.fuse_next = 0;
    norm_fwd(ex, st + o.c2, 512LL * 40 * W2, 40LL * W2, (long long)B * 512 * 40 * W2, ns, normp(P, nullptr, 6, 7, 10, 11), st + o.s2,
             st + o.y2, 256LL * 40 * W2, 40LL * W2, W2, nullptr, B, 256, 40, W2, ACT_GLU);
    // ---- :246  downSample2; output written straight in trunk layout [c*20+h][b][w]  (:249-251)
    ex.fuse_next = fuse_wino_norm();
    conv_fwd(ex, g.ds2, packed, B, 40, W2, CView{st + o.y2, 256LL * 40 * W2, 40LL * W2, W2}, View{st + o.c3, 512LL * 20 * W4, 20LL * W4, W4},
             (long long)B * 512 * 20 * W4, 0, 1, &ns);
    ex.fuse_next = 0;
    norm_fwd(ex, st + o.c3, 512LL * 20 * W4, 20LL * W4, (long long)B * 512 * 20 * W4, ns, normp(P, nullptr, 14, 15, 18, 19), st + o.s3,
             st + o.y3, W4, 20LL * BT4, (int)BT4, nullptr, B, 256, 20, W4, ACT_GLU);
    // ---- :254-255  1x1 5120->256 + IN ; image = [5120][B rows][W4]
    if (trunk_fwd_ksplit(ex, g.c2d1d, P, st + o.y3, st + o.c4, B, W4, &ns)) {}
    else conv_fwd(ex, g.c2d1d, packed, 1, B, W4, CView{st + o.y3, 0, BT4, W4}, View{st + o.c4, 0, BT4, W4}, 256 * BT4, 0, 1, &ns);
    norm_fwd(ex, st + o.c4, W4, BT4, 256 * BT4, ns, normp(P, nullptr, 22, 23), st + o.s4, st + o.y4, W4, BT4, W4, nullptr, B, 256, 1, W4, ACT_NONE);
    // ---- :258-271  six residual GLU blocks + 1x1 256->5120 + IN (written as NCHW [B][256][20][W4], 5120 = c*20 + h)
    const float* h = st + o.y4;
    if (trunk_net_enabled() && trunk_enabled() && ex.sync && mcvc_trunk_net_applies(B, W4)) {
        // small batch: the 12 dependent residual layers in ONE persistent launch (trunk.h)
        if (!ex.dry) {
            TrunkFwdNetArgs na{};
            int l = 0;
            auto fill = [&](const ConvSpec& c, int g0, int be0, int g1, int be1, const float* x, float* cvo, float* stats, float* y,
                            long long y_sn, long long y_sc, const float* res, int rows) {
                TrunkLayerDesc& d = na.L[l++];
                d.a0 = P[c.wi[0]]; d.bias0 = P[c.bi[0]]; d.gamma0 = P[g0]; d.beta0 = P[be0];
                if (c.nbr == 2) { d.a1 = P[c.wi[1]]; d.bias1 = P[c.bi[1]]; d.gamma1 = P[g1]; d.beta1 = P[be1]; }
                d.x = x; d.conv_out = cvo; d.stats = stats; d.y = y; d.y_sn = y_sn; d.y_sc = y_sc; d.res = res;
                d.Cin = c.Cin; d.KW = c.KW; d.M = c.Cout; d.mode = (c.nbr == 2) ? TRUNK_IN_GLU : TRUNK_IN; d.rows = rows;
            };
            for (int i = 0; i < 6; ++i) {
                const int b = 24 + 12 * i;
                fill(g.res_vg[i], b + 2, b + 3, b + 6, b + 7, h, st + o.r[i].ca, st + o.r[i].sa, st + o.r[i].ya, W4, BT4, nullptr, 8);
                fill(g.res_out[i], b + 10, b + 11, -1, -1, st + o.r[i].ya, st + o.r[i].cb, st + o.r[i].sb, st + o.r[i].y, W4, BT4, h, 4);
                h = st + o.r[i].y;
            }
            na.nlayers = l; na.B = B; na.T4 = W4; na.eps = kInEps; na.sync = ex.sync;
            ex.fail(mcvc_trunk_fwd_net_launch(na, ex.s));
        } else {
            h = st + o.r[5].y;
        }
        // 1x1 256 -> 5120 + IN: 320 independent row tiles -- its own (wide) launch
        if (!trunk_fwd(ex, g.c1d2d, P, 98, 99, -1, -1, h, st + o.c6, st + o.s6, st + o.y6, 5120LL * W4, W4, nullptr, B, W4)) ex.fail(MCVC_ERR_INVALID);
    } else {
    for (int i = 0; i < 6; ++i) {
        const int b = 24 + 12 * i;
        if (!trunk_fwd(ex, g.res_vg[i], P, b + 2, b + 3, b + 6, b + 7, h, st + o.r[i].ca, st + o.r[i].sa, st + o.r[i].ya, W4, BT4, nullptr, B, W4)) {
            conv_fwd(ex, g.res_vg[i], packed, 1, B, W4, CView{h, 0, BT4, W4}, View{st + o.r[i].ca, 0, BT4, W4}, 1024 * BT4, 0, 1, &ns);
            norm_fwd(ex, st + o.r[i].ca, W4, BT4, 1024 * BT4, ns, normp(P, nullptr, b + 2, b + 3, b + 6, b + 7), st + o.r[i].sa,
                     st + o.r[i].ya, W4, BT4, W4, nullptr, B, 512, 1, W4, ACT_GLU);
        }
        if (!trunk_fwd(ex, g.res_out[i], P, b + 10, b + 11, -1, -1, st + o.r[i].ya, st + o.r[i].cb, st + o.r[i].sb, st + o.r[i].y, W4, BT4, h, B, W4)) {
            conv_fwd(ex, g.res_out[i], packed, 1, B, W4, CView{st + o.r[i].ya, 0, BT4, W4}, View{st + o.r[i].cb, 0, BT4, W4}, 256 * BT4, 0, 1, &ns);
            norm_fwd(ex, st + o.r[i].cb, W4, BT4, 256 * BT4, ns, normp(P, nullptr, b + 10, b + 11), st + o.r[i].sb,
                     st + o.r[i].y, W4, BT4, W4, h, B, 256, 1, W4, ACT_NONE);
        }
        h = st + o.r[i].y;
    }
    // ---- :266-271  1x1 256->5120 + IN, written as NCHW [B][256][20][W4] (5120 = c*20 + h)
    if (!trunk_fwd(ex, g.c1d2d, P, 98, 99, -1, -1, h, st + o.c6, st + o.s6, st + o.y6, 5120LL * W4, W4, nullptr, B, W4)) {
        conv_fwd(ex, g.c1d2d, packed, 1, B, W4, CView{h, 0, BT4, W4}, View{st + o.c6, 0, BT4, W4}, 5120 * BT4, 0, 1, &ns);
        norm_fwd(ex, st + o.c6, W4, BT4, 5120 * BT4, ns, normp(P, nullptr, 98, 99), st + o.s6, st + o.y6, 5120LL * W4, W4, W4, nullptr,
                 B, 5120, 1, W4, ACT_NONE);
    }
    }
    // ---- :274  upSample1: conv 5x5 -> PixelShuffle(2) (fused into the store) -> IN -> x*sigmoid(x)
    ex.fuse_next = fuse_wino_norm();
    conv_fwd(ex, g.up1, packed, B, 20, W4, CView{st + o.y6, 256LL * 20 * W4, 20LL * W4, W4}, View{st + o.c7, 256LL * 40 * Wu1, 40LL * Wu1, Wu1},
             (long long)B * 256 * 40 * Wu1, 1, 1, &ns);
    ex.fuse_next = 0;
    norm_fwd(ex, st + o.c7, 256LL * 40 * Wu1, 40LL * Wu1, (long long)B * 256 * 40 * Wu1, ns, normp(P, nullptr, 106, 107), st + o.s7,
             st + o.y7, 256LL * 40 * Wu1, 40LL * Wu1, Wu1, nullptr, B, 256, 40, Wu1, ACT_SILU);
    // ---- :275  upSample2
    ex.fuse_next = fuse_wino_norm();
    conv_fwd(ex, g.up2, packed, B, 40, Wu1, CView{st + o.y7, 256LL * 40 * Wu1, 40LL * Wu1, Wu1}, View{st + o.c8, 128LL * 80 * Wu2, 80LL * Wu2, Wu2},
             (long long)B * 128 * 80 * Wu2, 1, 1, &ns);
    ex.fuse_next = 0;
    norm_fwd(ex, st + o.c8, 128LL * 80 * Wu2, 80LL * Wu2, (long long)B * 128 * 80 * Wu2, ns, normp(P, nullptr, 102, 103), st + o.s8,
             st + o.y8, 128LL * 80 * Wu2, 80LL * Wu2, Wu2, nullptr, B, 128, 80, Wu2, ACT_SILU);
    // ---- :278-279  last 5x15 conv to one channel
    conv_fwd(ex, g.last, packed, B, 80, Wu2, CView{st + o.y8, 128LL * 80 * Wu2, 80LL * Wu2, Wu2}, View{out, 80LL * Wu2, 80LL * Wu2, Wu2},
             (long long)B * 80 * Wu2, 0, 1, &ns);
    if (ns > 1) act_fwd(ex, out, (long long)B * 80 * Wu2, ns, nullptr, B, 1, 80 * Wu2, ACT_NONE);
}

// every gradient kernel launched so far (main stream: norm / bias gradients; aux stream: weight gradients) is ordered
// before `ev`
static void record_milestone(Exec& ex, void* ev)
{
    if (!ev || ex.dry) return;
    hipStream_t on = ex.s;
    if (ex.s2) {
        hipEvent_t e = pool_event();
        ex.fail(mcvc_event_record(e, ex.s));
        ex.fail(mcvc_stream_wait(ex.s2, e));
        on = ex.s2;
    }
    ex.fail(mcvc_event_record((hipEvent_t)ev, on));
}

// sample b0 of every stash tensor becomes sample 0: batch-major tensors move by b0 samples, the trunk-layout ones ([C][stash_B][W4]) by b0 rows
static void shift_stash(GenStash& s, const GenDims& d, int b0)
{
    if (b0 <= 0) return;
    const long long b = b0, T = d.T, W2 = d.W2, W4 = d.W4;
    s.xin += b * 2 * 80 * T; s.c1 += b * 256 * 80 * T; s.y1 += b * 128 * 80 * T;
    s.c2 += b * 512 * 40 * W2; s.s2 += b * 512 * 2; s.y2 += b * 256 * 40 * W2;
    s.c3 += b * 512 * 20 * W4; s.s3 += b * 512 * 2; s.y3 += b * W4;
    s.c4 += b * W4; s.s4 += b * 256 * 2; s.y4 += b * W4;
    for (int i = 0; i < 6; ++i) {
        s.r[i].ca += b * W4; s.r[i].sa += b * 1024 * 2; s.r[i].ya += b * W4;
        s.r[i].cb += b * W4; s.r[i].sb += b * 256 * 2; s.r[i].y += b * W4;
    }
    s.c6 += b * W4; s.s6 += b * 5120 * 2; s.y6 += b * 5120 * W4;
    s.c7 += b * 256 * 40 * d.Wu1; s.s7 += b * 256 * 2; s.y7 += b * 256 * 40 * d.Wu1;
    s.c8 += b * 128 * 80 * d.Wu2; s.s8 += b * 128 * 2; s.y8 += b * 128 * 80 * d.Wu2;
}

// `stash_B` > B: the stash was written by a forward pass over stash_B samples and this pass back-propagates through its samples
// [stash_b0, stash_b0 + B) only (the trainer's merged forwards: train.py:203-210 of iteration t+1 batched with :259-273 of iteration t, which
// need no gradient; and the identity sample's backward, which depends on nothing but its own forward, ahead of the others').  The 2-D
// tensors of the stash are batch-major, so a window is a pointer offset; the 1-D trunk's are [C][stash_B][W4]: their channel pitch is SBT4.
static void gen_backward_impl(Exec& ex, const float* const* P, const float* packed, float* const* G, const float* mask, const float* dout,
                              float* dx, int accumulate_dx, const float* stc, float* sc, const GenDims& d, void* const* milestones = nullptr,
                              int stash_B = 0, int stash_b0 = 0)
{
    ex.params = P;
    const GenNet& g = gen_net();
    if (stash_B < d.B) stash_B = d.B;
    GenStash o = gen_stash(gen_dims(stash_B, d.T));
    shift_stash(o, d, stash_b0);
    const GenScratch q = gen_scratch(d);
    float* st = const_cast<float*>(stc);     // stash is read-only here; kernels take non-const for slab-reduce paths that are not used on it
    const int B = d.B, T = d.T, W2 = d.W2, W4 = d.W4, Wu1 = d.Wu1, Wu2 = d.Wu2;
    const long long BT4 = (long long)B * W4;
    const long long SBT4 = (long long)stash_B * W4;
    const int pxB = stash_B > B ? stash_B : 0;
    // (float4 paths of the trunk kernels need 16-byte aligned rows: a window that starts at an odd multiple of W4 < 4 floats would not be)
    if (stash_b0 > 0 && (((long long)stash_b0 * W4) & 3)) { ex.fail(MCVC_ERR_INVALID); return; }
    // dY buffers alternate between two copies so an aux-stream weight gradient can still read layer k's dY while the
    // main stream already produces layer k-1's
    float* GA = sc + q.ga;
    float* GBs[2] = {sc + q.gb, sc + q.gb2};
    float* DT1s[2] = {sc + q.dt1, sc + q.dt1b};
    float* DT3s[2] = {sc + q.dt3, sc + q.dt3b};
    int gbi = 0;
    auto nextGB = [&]() { gbi ^= 1; return GBs[gbi]; };
    float* GB = GBs[0];
    float* DH = sc + q.dh; float* DT1 = DT1s[0]; float* DT2 = sc + q.dt2; float* DT3 = DT3s[0];
    int ns = 1;
    // ---- last conv (model.py:278)
    {
        CView dyv{dout, 80LL * Wu2, 80LL * Wu2, Wu2};
        CView xv{st + o.y8, 128LL * 80 * Wu2, 80LL * Wu2, Wu2};
        conv_wgrad(ex, g.last, G, B, 80, Wu2, xv, dyv);
        conv_bias_grad(ex, g.last, G, B, dyv, 80 * Wu2);
        conv_dgrad(ex, g.last, packed, B, 80, Wu2, dyv, View{GA, 128LL * 80 * Wu2, 80LL * Wu2, Wu2}, (long long)B * 128 * 80 * Wu2, 0, 1, &ns);
    }
    // ---- upSample2 (:275): IN+SiLU backward, un-shuffled into conv-output coordinates [B][512][40][Wu1]
    GB = nextGB();
    norm_bwd(ex, st + o.c8, 128LL * 80 * Wu2, 80LL * Wu2, normp(P, G, 102, 103), st + o.s8, GA, 128LL * 80 * Wu2, 80LL * Wu2, Wu2,
             (long long)B * 128 * 80 * Wu2, ns, GB, 512LL * 40 * Wu1, 40LL * Wu1, Wu1, 1, B, 128, 80, Wu2, ACT_SILU);
    {
        CView dyv{GB, 512LL * 40 * Wu1, 40LL * Wu1, Wu1};
        CView xv{st + o.y7, 256LL * 40 * Wu1, 40LL * Wu1, Wu1};
        conv_wgrad(ex, g.up2, G, B, 40, Wu1, xv, dyv);
        conv_bias_grad(ex, g.up2, G, B, dyv, 40 * Wu1);
        conv_dgrad(ex, g.up2, packed, B, 40, Wu1, dyv, View{GA, 256LL * 40 * Wu1, 40LL * Wu1, Wu1}, (long long)B * 256 * 40 * Wu1, 0, 1, &ns);
    }
    // ---- upSample1 (:274)
    GB = nextGB();
    norm_bwd(ex, st + o.c7, 256LL * 40 * Wu1, 40LL * Wu1, normp(P, G, 106, 107), st + o.s7, GA, 256LL * 40 * Wu1, 40LL * Wu1, Wu1,
             (long long)B * 256 * 40 * Wu1, ns, GB, 1024LL * 20 * W4, 20LL * W4, W4, 1, B, 256, 40, Wu1, ACT_SILU);
    {
        CView dyv{GB, 1024LL * 20 * W4, 20LL * W4, W4};
        CView xv{st + o.y6, 256LL * 20 * W4, 20LL * W4, W4};
        conv_wgrad(ex, g.up1, G, B, 20, W4, xv, dyv);
        conv_bias_grad(ex, g.up1, G, B, dyv, 20 * W4);
        conv_dgrad(ex, g.up1, packed, B, 20, W4, dyv, View{GA, 256LL * 20 * W4, 20LL * W4, W4}, (long long)B * 5120 * W4, 0, 1, &ns);
    }
    if (milestones) record_milestone(ex, milestones[0]);          // parameters [100,110) are done
    // ---- conv1dto2d + IN (:266-271): dy arrives NCHW, dx leaves in trunk layout
    GB = nextGB();
    norm_bwd(ex, st + o.c6, W4, SBT4, normp(P, G, 98, 99), st + o.s6, GA, 5120LL * W4, W4, W4, 5120 * BT4, ns,
             GB, W4, BT4, W4, 0, B, 5120, 1, W4, ACT_NONE);
    {
        const float* hin = st + o.r[5].y;
        CView dyv{GB, 0, BT4, W4};
        conv_wgrad(ex, g.c1d2d, G, 1, B, W4, CView{hin, 0, SBT4, W4}, dyv);
        if (trunk_dgrad(ex, g.c1d2d, packed, GB, DH, 0, B, W4, &ns)) {}
        else conv_dgrad(ex, g.c1d2d, packed, 1, B, W4, dyv, View{DH, 0, BT4, W4}, 256 * BT4, 0, 1, &ns);
    }
    // ---- residual blocks (:258-263), last to first.  DH carries d(h) and is updated in place.
    // small batch: the InstanceNorm backward of each layer is recomputed inside the data-gradient launch that consumes it (one thread
    // per channel of the workgroup's K slice) -- 2 dependent launches per residual block instead of 4.  Needs an un-split upstream
    // gradient (the fused staging does not sum slabs) and the fused data-gradient kernels for both convs of the block.
    static const int fuse_knob = mcvc_knob("MCVC_TRUNK_BWD_FUSE", 1);
    static const int batch_knob = mcvc_knob("MCVC_TRUNK_WGRAD_BATCH", 1);
    SmallKJob wjobs[MCVC_SMALLK_MAX_JOBS];
    int nwjobs = 0;
    const bool batch_w = batch_knob && mcvc_wgrad_smallk_batch_applies(B, W4) && (W4 <= 128);
    // Persistent backward (trunk.h): the 12 dependent data-gradient layers of the six blocks in ONE launch, when every layer would take the
    // fused path anyway, the gradient arriving from conv1dto2d is not split into slabs and the staged gradients fit the LDS
    static const int bwd_net_knob = mcvc_knob("MCVC_TRUNK_BWD_NET", 1);
    bool bwd_net = bwd_net_knob && trunk_bwd_net_enabled() && fuse_knob && trunk_enabled() && batch_w && !ex.dry && G && ex.sync &&
                   mcvc_trunk_bwd_net_applies(B, W4);
    for (int i = 0; i < 6 && bwd_net; ++i)
        bwd_net = g.res_out[i].off_tk >= 0 && g.res_vg[i].off_tk >= 0 && mcvc_trunk_applies(g.res_out[i].cout_tot, 3, g.res_out[i].Cin, B, W4, TRUNK_PLAIN, 1);
    if (bwd_net) {
        TrunkBwdNetArgs na{};
        int l = 0;
        for (int i = 5; i >= 0; --i) {
            const int b = 24 + 12 * i;
            const float* hin = (i == 0) ? (st + o.y4) : (st + o.r[i - 1].y);
            float* dt3 = sc + q.dtx3 + (long long)i * 256 * BT4;
            float* dt1 = sc + q.dtx1 + (long long)i * 1024 * BT4;
            TrunkBwdLayerDesc& da = na.L[l++];           // conv1d_out_layer (512 -> 256) + its InstanceNorm: d(h_out) -> d(GLU output)
            da.wt = packed + g.res_out[i].off_tk; da.dy = DH; da.px = st + o.r[i].cb; da.stats = st + o.r[i].sb;
            da.g0 = P[b + 10]; da.b0 = P[b + 11]; da.xout = dt3; da.pxB = pxB; da.dg0 = G[b + 10]; da.db0 = G[b + 11];
            da.out = DT2; da.pre = 1; da.C = 256; da.M = 512; da.rows = 8;
            da.flags = (i < 5 ? TBWD_DY_FRESH : 0) | ((i == 5 && ns > 1) ? TBWD_SLAB_DY : 0);            // (conv1dto2d's K-split slabs)
            TrunkBwdLayerDesc& db = na.L[l++];           // value | gate convs (256 -> 512 each) + norms + GLU: d(GLU output) -> d(h_in), added to the skip path
            db.wt = packed + g.res_vg[i].off_tk; db.dy = DT2; db.px = st + o.r[i].ca; db.stats = st + o.r[i].sa;
            db.g0 = P[b + 2]; db.b0 = P[b + 3]; db.g1 = P[b + 6]; db.b1 = P[b + 7]; db.xout = dt1; db.pxB = pxB;
            db.dg0 = G[b + 2]; db.db0 = G[b + 3]; db.dg1 = G[b + 6]; db.db1 = G[b + 7];
            db.out = DH; db.pre = 2; db.C = 512; db.M = 256; db.rows = 4;
            db.flags = TBWD_ACCUMULATE | TBWD_DY_FRESH | ((i == 5 && ns > 1) ? TBWD_SLAB_OUT : 0);
            wjobs[nwjobs++] = SmallKJob{st + o.r[i].ya, dt3, G[g.res_out[i].wi[0]], 512, 256, pxB};
            wjobs[nwjobs++] = SmallKJob{hin, dt1, G[g.res_vg[i].wi[0]], 256, 512, pxB};
            wjobs[nwjobs++] = SmallKJob{hin, dt1 + 512LL * BT4, G[g.res_vg[i].wi[1]], 256, 512, pxB};
        }
        na.nlayers = l; na.B = B; na.T4 = W4; na.slabs = ex.slabs; na.slab_stride = 256 * BT4; na.nslab = ns; na.sync = ex.sync + MCVC_TRUNK_SYNC_WORDS; na.err = ex.sync + MCVC_TRUNK_SYNC_WORDS - 1;
        for (int i = 0; i < 6; ++i) { wait_readers(ex, sc + q.dtx3 + (long long)i * 256 * BT4); wait_readers(ex, sc + q.dtx1 + (long long)i * 1024 * BT4); }
        ex.fail(mcvc_trunk_bwd_net_launch(na, ex.s));
        ns = 1;
    }
    for (int i = 5; i >= 0 && !bwd_net; --i) {
        const int b = 24 + 12 * i;
        const float* hin = (i == 0) ? (st + o.y4) : (st + o.r[i - 1].y);
        DT3 = DT3s[i & 1]; DT1 = DT1s[i & 1];
        const bool fuse = fuse_knob && trunk_enabled() && ns == 1 && !ex.dry && G && g.res_out[i].off_tk >= 0 && g.res_vg[i].off_tk >= 0 &&
                          mcvc_trunk_applies(g.res_out[i].cout_tot, 3, g.res_out[i].Cin, B, W4, TRUNK_PLAIN, 1) &&
                          trunk_pick_ksplit(g.res_vg[i].cout_tot, 3, g.res_vg[i].Cin, B, W4, 0) >= 1;
        if (fuse) {
            if (batch_w) { DT3 = sc + q.dtx3 + (long long)i * 256 * BT4; DT1 = sc + q.dtx1 + (long long)i * 1024 * BT4; }
            TrunkPre pa{1, st + o.r[i].cb, st + o.r[i].sb, P[b + 10], P[b + 11], nullptr, nullptr, DT3, G[b + 10], G[b + 11], nullptr, nullptr, pxB};
            trunk_dgrad(ex, g.res_out[i], packed, DH, DT2, 0, B, W4, nullptr, &pa);
            TrunkPre pb{2, st + o.r[i].ca, st + o.r[i].sa, P[b + 2], P[b + 3], P[b + 6], P[b + 7], DT1, G[b + 2], G[b + 3], G[b + 6], G[b + 7], pxB};
            ns = 1;
            trunk_dgrad(ex, g.res_vg[i], packed, DT2, DH, 1, B, W4, &ns, &pb);
            if (batch_w) {          // weight gradients of the block: queued for the batched launch behind the chain
                wjobs[nwjobs++] = SmallKJob{st + o.r[i].ya, DT3, G[g.res_out[i].wi[0]], 512, 256, pxB};
                wjobs[nwjobs++] = SmallKJob{hin, DT1, G[g.res_vg[i].wi[0]], 256, 512, pxB};
                wjobs[nwjobs++] = SmallKJob{hin, DT1 + 512LL * BT4, G[g.res_vg[i].wi[1]], 256, 512, pxB};
            } else {
                conv_wgrad(ex, g.res_out[i], G, 1, B, W4, CView{st + o.r[i].ya, 0, SBT4, W4}, CView{DT3, 0, BT4, W4});
                conv_wgrad(ex, g.res_vg[i], G, 1, B, W4, CView{hin, 0, SBT4, W4}, CView{DT1, 0, BT4, W4});
            }
            continue;
        }
        norm_bwd(ex, st + o.r[i].cb, W4, SBT4, normp(P, G, b + 10, b + 11), st + o.r[i].sb, DH, W4, BT4, W4, 256 * BT4, ns,
                 DT3, W4, BT4, W4, 0, B, 256, 1, W4, ACT_NONE);
        int ns2 = 1;
        {
            CView dyv{DT3, 0, BT4, W4};
            conv_wgrad(ex, g.res_out[i], G, 1, B, W4, CView{st + o.r[i].ya, 0, SBT4, W4}, dyv);
            if (!trunk_dgrad(ex, g.res_out[i], packed, DT3, DT2, 0, B, W4))
                conv_dgrad(ex, g.res_out[i], packed, 1, B, W4, dyv, View{DT2, 0, BT4, W4}, 512 * BT4, 0, 1, &ns2);
        }
        norm_bwd(ex, st + o.r[i].ca, W4, SBT4, normp(P, G, b + 2, b + 3, b + 6, b + 7), st + o.r[i].sa, DT2, W4, BT4, W4, 512 * BT4, ns2,
                 DT1, W4, BT4, W4, 0, B, 512, 1, W4, ACT_GLU);
        {
            CView dyv{DT1, 0, BT4, W4};
            conv_wgrad(ex, g.res_vg[i], G, 1, B, W4, CView{hin, 0, SBT4, W4}, dyv);
            ns = 1;          // (deterministic mode: DH is left alone and its K-split partials go to slabs the next norm_bwd sums)
            if (!trunk_dgrad(ex, g.res_vg[i], packed, DT1, DH, 1 /*accumulate: skip path*/, B, W4, &ns))
                conv_dgrad(ex, g.res_vg[i], packed, 1, B, W4, dyv, View{DH, 0, BT4, W4}, 256 * BT4, 1, 1, &ns);      // (a K split leaves its partials in slabs the next norm_bwd sums)
        }
    }
    if (nwjobs > 0) {              // ONE launch for the residual layers' weight gradients, beside the rest of the backward chain
        hipStream_t ws = ex.s;
        if (ex.s2) {
            hipEvent_t e = pool_event();
            ex.fail(mcvc_event_record(e, ex.s));
            ex.fail(mcvc_stream_wait(ex.s2, e));
            ws = ex.s2;
        }
        ex.fail(mcvc_wgrad_smallk_batch_launch(wjobs, nwjobs, B, W4, ws));
    }
    if (milestones) record_milestone(ex, milestones[1]);          // parameters [24,100) are done
    // ---- conv2dto1d + IN (:254-255)
    DT3 = DT3s[1];
    norm_bwd(ex, st + o.c4, W4, SBT4, normp(P, G, 22, 23), st + o.s4, DH, W4, BT4, W4, 256 * BT4, ns, DT3, W4, BT4, W4, 0, B, 256, 1, W4, ACT_NONE);
    {
        CView dyv{DT3, 0, BT4, W4};
        conv_wgrad(ex, g.c2d1d, G, 1, B, W4, CView{st + o.y3, 0, SBT4, W4}, dyv);
        ns = 1;
        if (!trunk_dgrad(ex, g.c2d1d, packed, DT3, GA, 0, B, W4))
            conv_dgrad(ex, g.c2d1d, packed, 1, B, W4, dyv, View{GA, 0, BT4, W4}, 5120 * BT4, 0, 1, &ns);
    }
    // ---- downSample2 (:246): dy is in trunk layout
    GB = nextGB();
    norm_bwd(ex, st + o.c3, 512LL * 20 * W4, 20LL * W4, normp(P, G, 14, 15, 18, 19), st + o.s3, GA, W4, 20LL * BT4, (int)BT4, 5120 * BT4, ns,
             GB, 512LL * 20 * W4, 20LL * W4, W4, 0, B, 256, 20, W4, ACT_GLU);
    {
        CView dyv{GB, 512LL * 20 * W4, 20LL * W4, W4};
        conv_wgrad(ex, g.ds2, G, B, 40, W2, CView{st + o.y2, 256LL * 40 * W2, 40LL * W2, W2}, dyv);
        conv_dgrad(ex, g.ds2, packed, B, 40, W2, dyv, View{GA, 256LL * 40 * W2, 40LL * W2, W2}, (long long)B * 256 * 40 * W2, 0, 1, &ns);
    }
    if (milestones && ex.fine_ms) record_milestone(ex, milestones[2]);          // parameters [12,24) (downSample2, conv2dto1d) are done
    // ---- downSample1 (:245)
    GB = nextGB();
    norm_bwd(ex, st + o.c2, 512LL * 40 * W2, 40LL * W2, normp(P, G, 6, 7, 10, 11), st + o.s2, GA, 256LL * 40 * W2, 40LL * W2, W2,
             (long long)B * 256 * 40 * W2, ns, GB, 512LL * 40 * W2, 40LL * W2, W2, 0, B, 256, 40, W2, ACT_GLU);
    {
        CView dyv{GB, 512LL * 40 * W2, 40LL * W2, W2};
        conv_wgrad(ex, g.ds1, G, B, 80, T, CView{st + o.y1, 128LL * 80 * T, 80LL * T, T}, dyv);
        conv_dgrad(ex, g.ds1, packed, B, 80, T, dyv, View{GA, 128LL * 80 * T, 80LL * T, T}, (long long)B * 128 * 80 * T, 0, 1, &ns);
    }
    if (milestones && ex.fine_ms) record_milestone(ex, milestones[3]);          // parameters [4,12) (downSample1) are done
    // ---- conv1 gated GLU (:242)
    GB = nextGB();
    act_bwd(ex, st + o.c1, GA, (long long)B * 128 * 80 * T, ns, GB, B, 128, 80 * T, ACT_GLU);
    {
        CView dyv{GB, 256LL * 80 * T, 80LL * T, T};
        ex.br1_on_main = (dx == nullptr) ? 1 : 0;
        conv_wgrad(ex, g.conv1, G, B, 80, T, CView{st + o.xin, 2LL * 80 * T, 80LL * T, T}, dyv);
        ex.br1_on_main = 0;
        conv_bias_grad(ex, g.conv1, G, B, dyv, 80 * T);
        if (dx) {
            conv_dgrad(ex, g.conv1, packed, B, 80, T, dyv, View{GA, 2LL * 80 * T, 80LL * T, T}, (long long)B * 2 * 80 * T, 0, 1, &ns);
            if (!ex.dry) ex.fail(mcvc_mask_grad_launch(GA, ex.slabs, (long long)B * 2 * 80 * T, ns, mask, dx, B, 80 * T, 2, accumulate_dx, ex.s));
        }
    }
    if (ex.no_join) ex.readers.clear(); else join_aux(ex);
}

// =================================================================================================
// Discriminator
// =================================================================================================
struct DiscNet { ConvSpec conv1, ds[3], outc; long long packed_floats; };
static DiscNet build_disc()
{
    DiscNet n{};
    long long cur = 0;
    n.conv1 = mk(1, 128, 1, 3, 3, 1, 1, 1, 0, 1, -1, -1, 1);             // model.py:290-295
    n.ds[0] = mk(128, 256, 1, 3, 3, 2, 1, 1, 2, 3, -1, -1, 0);           // :298-302
    n.ds[1] = mk(256, 512, 1, 3, 3, 2, 1, 1, 6, 7, -1, -1, 0);           // :304-308
    n.ds[2] = mk(512, 1024, 1, 3, 3, 2, 1, 1, 10, 11, -1, -1, 0);        // :310-314
    n.outc = mk(1024, 1, 1, 1, 3, 1, 0, 1, 18, 19, -1, -1, 1);           // :323-327   (downSample4, params 14-17, is dead: :316-320 never called)
    spec_finalize(n.conv1, cur);
    for (int i = 0; i < 3; ++i) spec_finalize(n.ds[i], cur);
    spec_finalize(n.outc, cur);
    n.packed_floats = cur;
    return n;
}
static const DiscNet& disc_net() { static const DiscNet n = build_disc(); return n; }

struct DiscDims { int B, T, H[4], W[4]; long long big; };
static DiscDims disc_dims(int B, int T)
{
    DiscDims d{};
    d.B = B; d.T = T; d.H[0] = 80; d.W[0] = T;
    for (int i = 1; i < 4; ++i) { d.H[i] = conv_out(d.H[i - 1], 3, 2, 1); d.W[i] = conv_out(d.W[i - 1], 3, 2, 1); }
    d.big = (long long)B * 128 * 80 * T;
    // (gradient ping-pong buffers also hold dY in the padded layout of the implicit data gradient: (OH + 1) x (OW + 4) per plane)
    for (int i = 1; i < 4; ++i) { const long long p = (long long)B * (128 << i) * mcvc_dyp_plane(d.H[i], d.W[i]); if (p > d.big) d.big = p; }
    return d;
}
static const int kDC[4] = {128, 256, 512, 1024};

struct DiscStash { long long c0, y0, c[3], s[3], y[3], logit, total; };
static DiscStash disc_stash(const DiscDims& d)
{
    DiscStash s{};
    long long cur = 0;
    auto take = [&](long long n) { const long long o = cur; cur += (n + 3) & ~3LL; return o; };
    const long long B = d.B;
    // (the inputs of the three stride-2 layers -- y0, y[0], y[1] -- may be stored in the phase-split padded layout of sgemm.h: the larger size)
    auto xs_or_dense = [&](int C, int H, int W) { const long long xs = (H & 1) || (W & 1) ? 0 : mcvc_xs_floats(C, H, W); const long long dn = (long long)C * H * W; return B * (xs > dn ? xs : dn); };
    s.c0 = take(B * 128 * 80 * d.T); s.y0 = take(xs_or_dense(128, 80, d.T));
    for (int i = 0; i < 3; ++i) {
        const long long n = B * kDC[i + 1] * d.H[i + 1] * d.W[i + 1];
        s.c[i] = take(n); s.s[i] = take(B * kDC[i + 1] * 2); s.y[i] = take(i < 2 ? xs_or_dense(kDC[i + 1], d.H[i + 1], d.W[i + 1]) : n);
    }
    s.logit = take(B * d.H[3] * d.W[3]);
    s.total = cur;
    return s;
}
struct DiscScratch { long long ga, gb, gb2, slabs; };
static DiscScratch disc_scratch(const DiscDims& d)
{
    DiscScratch s{};
    s.ga = 0; s.gb = (d.big + 3) & ~3LL; s.gb2 = 2 * s.gb; s.slabs = 3 * s.gb;
    return s;
}

// the first (one input channel) and the output (one output channel) layer on their own kernels (fewout_kernels.hip) instead of the generic
// conv + activation pairs
static int disc_out_direct()
{
    static const int en = mcvc_knob("MCVC_DISC_OUT", 1);
    return en;
}

// every stride-2 layer of a (B, T) pass takes the implicit-GEMM kernels (and the first layer its own kernel, which stores the phase-split layout)
static bool disc_igemm(const DiscDims& d)
{
    const DiscNet& n = disc_net();
    bool ok = disc_out_direct() != 0;
    for (int i = 0; i < 3 && ok; ++i) ok = igemm_applies(n.ds[i], d.H[i], d.W[i]);
    return ok;
}

static void disc_forward_impl(Exec& ex, const float* const* P, const float* packed, const float* x, float* out, float* st, const DiscDims& d)
{
    ex.params = P;
    const DiscNet& n = disc_net();
    const DiscStash o = disc_stash(d);
    const int B = d.B, T = d.T;
    int ns = 1;
    const bool ig = disc_igemm(d);           // the three stride-2 layers as implicit GEMMs: their inputs are stored phase-split (sgemm.h)
    // model.py:343-344  unsqueeze(1) -> conv 3x3 -> x*sigmoid(x)
    if (disc_out_direct() && !ex.dry) {
        ex.fail(mcvc_disc_conv1_fwd_launch(x, P[n.conv1.wi[0]], P[n.conv1.bi[0]], st + o.c0, st + o.y0, B, 128, 80, T, ex.s, ig ? 1 : 0));
    } else {
        conv_fwd(ex, n.conv1, packed, B, 80, T, CView{x, 80LL * T, 80LL * T, T}, View{st + o.c0, 128LL * 80 * T, 80LL * T, T},
                 (long long)B * 128 * 80 * T, 0, 1, &ns);
        act_fwd(ex, st + o.c0, (long long)B * 128 * 80 * T, ns, st + o.y0, B, 128, 80 * T, ACT_SILU);
    }
    const float* h = st + o.y0;
    for (int i = 0; i < 3; ++i) {                       // :345-347
        const int Ci = kDC[i], Co = kDC[i + 1], Hi = d.H[i], Wi = d.W[i], Ho = d.H[i + 1], Wo = d.W[i + 1];
        if (ig) conv_fwd_igemm(ex, n.ds[i], packed, B, Hi, Wi, h, View{st + o.c[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo},
                               (long long)B * Co * Ho * Wo, 1, &ns);
        else conv_fwd(ex, n.ds[i], packed, B, Hi, Wi, CView{h, (long long)Ci * Hi * Wi, (long long)Hi * Wi, Wi},
                      View{st + o.c[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo}, (long long)B * Co * Ho * Wo, 0, 1, &ns);
        if (ig && i < 2) {                              // this layer's output feeds the next stride-2 layer: phase-split padded store
            const long long pl = mcvc_xs_plane(Ho, Wo);
            ex.y_xs_pw = mcvc_xs_pw(Wo); ex.y_xs_plane = pl;
            norm_fwd(ex, st + o.c[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, (long long)B * Co * Ho * Wo, ns, normp(P, nullptr, 4 + 4 * i, 5 + 4 * i),
                     st + o.s[i], st + o.y[i], (long long)Co * 4 * pl, 4 * pl, Wo, nullptr, B, Co, Ho, Wo, ACT_SILU);
        } else
        norm_fwd(ex, st + o.c[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, (long long)B * Co * Ho * Wo, ns, normp(P, nullptr, 4 + 4 * i, 5 + 4 * i),
                 st + o.s[i], st + o.y[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo, nullptr, B, Co, Ho, Wo, ACT_SILU);
        h = st + o.y[i];
    }
    // :348  1x3 conv to one channel + sigmoid
    const int H3 = d.H[3], W3 = d.W[3];
    if (disc_out_direct() && !ex.dry) {
        ex.fail(mcvc_disc_out_fwd_launch(h, P[n.outc.wi[0]], P[n.outc.bi[0]], st + o.logit, out, B, 1024, H3, W3, ex.s));
        return;
    }
    conv_fwd(ex, n.outc, packed, B, H3, W3, CView{h, 1024LL * H3 * W3, (long long)H3 * W3, W3}, View{st + o.logit, (long long)H3 * W3, (long long)H3 * W3, W3},
             (long long)B * H3 * W3, 0, 1, &ns);
    act_fwd(ex, st + o.logit, (long long)B * H3 * W3, ns, out, B, 1, H3 * W3, ACT_SIGMOID);
}

static void disc_backward_impl(Exec& ex, const float* const* P, const float* packed, float* const* G, const float* dout, int is_logit_grad,
                               float* dx, int accumulate_dx, const float* stc, float* sc, const DiscDims& d)
{
    ex.params = P;
    const DiscNet& n = disc_net();
    const DiscStash o = disc_stash(d);
    const DiscScratch q = disc_scratch(d);
    float* st = const_cast<float*>(stc);
    const int B = d.B, T = d.T;
    float* GA = sc + q.ga;
    float* GBs[2] = {sc + q.gb, sc + q.gb2};
    int gbi = 0;
    auto nextGB = [&]() { gbi ^= 1; return GBs[gbi]; };
    float* GB = GBs[0];
    const int H3 = d.H[3], W3 = d.W[3];
    int ns = 1;
    const float* dlogit = dout;
    if (!is_logit_grad) {                         // sigmoid backward (model.py:348)
        if (!ex.dry) {
            ex.fail(mcvc_copy_launch(dout, GA, B * H3 * W3, ex.s));
        }
        act_bwd(ex, st + o.logit, GA, 0, 1, GB, B, 1, H3 * W3, ACT_SIGMOID);
        dlogit = GB;
    }
    {
        CView dyv{dlogit, (long long)H3 * W3, (long long)H3 * W3, W3};
        CView xv{st + o.y[2], 1024LL * H3 * W3, (long long)H3 * W3, W3};
        conv_wgrad(ex, n.outc, G, B, H3, W3, xv, dyv);
        conv_bias_grad(ex, n.outc, G, B, dyv, H3 * W3);
        if (disc_out_direct() && !ex.dry) ex.fail(mcvc_disc_out_dgrad_launch(dlogit, P[n.outc.wi[0]], GA, B, 1024, H3, W3, ex.s));
        else conv_dgrad(ex, n.outc, packed, B, H3, W3, dyv, View{GA, 1024LL * H3 * W3, (long long)H3 * W3, W3}, (long long)B * 1024 * H3 * W3, 0, 1, &ns);
    }
    const bool ig = disc_igemm(d);
    for (int i = 2; i >= 0; --i) {
        const int Ci = kDC[i], Co = kDC[i + 1], Hi = d.H[i], Wi = d.W[i], Ho = d.H[i + 1], Wo = d.W[i + 1];
        GB = nextGB();
        const float* hin = (i == 0) ? (st + o.y0) : (st + o.y[i - 1]);
        if (ig) {
            // dY leaves the InstanceNorm backward in the padded layout the implicit data gradient gathers from (zero column / row written there);
            // the layer's input is stored phase-split (forward): the weight gradient's staging reads both as they are
            const int pitch = mcvc_dyp_pitch(Wo);
            const long long pl = mcvc_dyp_plane(Ho, Wo), xpl = mcvc_xs_plane(Hi, Wi);
            ex.dx_pitch = pitch;
            norm_bwd(ex, st + o.c[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, normp(P, G, 4 + 4 * i, 5 + 4 * i), st + o.s[i],
                     GA, (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo, (long long)B * Co * Ho * Wo, ns,
                     GB, (long long)Co * pl, pl, pitch, 0, B, Co, Ho, Wo, ACT_SILU);
            ex.wgrad_x_xs = 1;
            conv_wgrad(ex, n.ds[i], G, B, Hi, Wi, CView{hin, (long long)Ci * 4 * xpl, 4 * xpl, Wi}, CView{GB, (long long)Co * pl, pl, pitch});
            ex.wgrad_x_xs = 0;
            conv_dgrad_igemm(ex, n.ds[i], packed, B, Hi, Wi, GB, View{GA, (long long)Ci * Hi * Wi, (long long)Hi * Wi, Wi}, (long long)B * Ci * Hi * Wi, 0, 1, &ns);
            continue;
        }
        norm_bwd(ex, st + o.c[i], (long long)Co * Ho * Wo, (long long)Ho * Wo, normp(P, G, 4 + 4 * i, 5 + 4 * i), st + o.s[i],
                 GA, (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo, (long long)B * Co * Ho * Wo, ns,
                 GB, (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo, 0, B, Co, Ho, Wo, ACT_SILU);
        CView dyv{GB, (long long)Co * Ho * Wo, (long long)Ho * Wo, Wo};
        conv_wgrad(ex, n.ds[i], G, B, Hi, Wi, CView{hin, (long long)Ci * Hi * Wi, (long long)Hi * Wi, Wi}, dyv);
        conv_dgrad(ex, n.ds[i], packed, B, Hi, Wi, dyv, View{GA, (long long)Ci * Hi * Wi, (long long)Hi * Wi, Wi}, (long long)B * Ci * Hi * Wi, 0, 1, &ns);
    }
    GB = nextGB();
    act_bwd(ex, st + o.c0, GA, (long long)B * 128 * 80 * T, ns, GB, B, 128, 80 * T, ACT_SILU);
    {
        CView dyv{GB, 128LL * 80 * T, 80LL * T, T};
        // the forward input x is not in the stash; its weight gradient needs it -> caller passes it via stash? no: keep a copy
        conv_wgrad(ex, n.conv1, G, B, 80, T, CView{st + o.total, 80LL * T, 80LL * T, T}, dyv);
        conv_bias_grad(ex, n.conv1, G, B, dyv, 80 * T);
        if (dx) {
            conv_dgrad(ex, n.conv1, packed, B, 80, T, dyv, View{GA, 80LL * T, 80LL * T, T}, (long long)B * 80 * T, 0, 1, &ns);
            if (!ex.dry) ex.fail(mcvc_mask_grad_launch(GA, ex.slabs, (long long)B * 80 * T, ns, nullptr, dx, B, 80 * T, 1, accumulate_dx, ex.s));
        }
    }
    join_aux(ex);
}

struct Needs { long long slab, wslab, sg, sgw; };

// split of the scratch tail into [conv slabs | wgrad slabs]: from a dry run of the schedule, cached per (net, B, T)
template <class F>
static Needs cached_needs(int kind, int B, int T, F&& dry_run)
{
    static std::mutex mu;
    static std::map<long long, Needs> cache;
    const long long key = ((long long)kind << 60) ^ ((long long)B << 32) ^ (long long)T;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    Exec ex{}; ex.dry = true;
    dry_run(ex);
    Needs n{(ex.slab_need + 3) & ~3LL, (ex.wslab_need + 3) & ~3LL, (ex.sg_need + 3) & ~3LL, (ex.sgw_need + 3) & ~3LL};
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = n;
    return n;
}

static Exec make_exec(void* stream, void* aux_stream, float* scratch, long long scratch_floats, long long slab_off, const Needs& nd)
{
    Exec ex{};
    ex.s = (hipStream_t)stream; ex.s2 = (hipStream_t)aux_stream; ex.dry = false; ex.err = 0;
    ex.slabs = scratch + slab_off;
    ex.slab_cap = nd.slab;
    ex.sg = nd.sg ? scratch + slab_off + nd.slab : nullptr;
    ex.sg_cap = nd.sg;
    ex.sgw = nd.sgw ? scratch + slab_off + nd.slab + nd.sg : nullptr;
    ex.sgw_cap = nd.sgw;
    ex.wslabs = scratch + slab_off + nd.slab + nd.sg + nd.sgw;
    ex.wslab_cap = scratch_floats - slab_off - nd.slab - nd.sg - nd.sgw;
    return ex;
}

static Needs gen_needs(int B, int T)
{
    return cached_needs(1, B, T, [&](Exec& ex) {
        const GenDims d = gen_dims(B, T);
        gen_forward_impl(ex, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d);
        float dummy = 0.f;
        gen_backward_impl(ex, nullptr, nullptr, nullptr, nullptr, nullptr, &dummy, 0, nullptr, nullptr, d);
    });
}

static Needs disc_needs(int B, int T)
{
    return cached_needs(2, B, T, [&](Exec& ex) {
        const DiscDims d = disc_dims(B, T);
        disc_forward_impl(ex, nullptr, nullptr, nullptr, nullptr, nullptr, d);
        float dummy = 0.f;
        disc_backward_impl(ex, nullptr, nullptr, nullptr, nullptr, 1, &dummy, 0, nullptr, nullptr, d);
    });
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int mcvc_version(void) { return MCVC_ABI_VERSION; }

// ---- grouped ("twin") launches (twin.h): the caller brackets two identical call sequences on different networks -------------------
static thread_local TwinCtx t_twin_ctx;
int mcvc_twin_begin(void)
{
    if (g_mcvc_twin) return MCVC_ERR_INVALID;
    t_twin_ctx.recs.clear(); t_twin_ctx.args.clear(); t_twin_ctx.next = 0; t_twin_ctx.err = 0; t_twin_ctx.phase = 1;
    g_mcvc_twin = &t_twin_ctx;
    return 0;
}
int mcvc_twin_switch(void)
{
    if (!g_mcvc_twin || g_mcvc_twin->phase != 1) return MCVC_ERR_INVALID;
    g_mcvc_twin->phase = 2;
    g_mcvc_twin->next = 0;
    return g_mcvc_twin->err;
}
int mcvc_twin_end(void)
{
    if (!g_mcvc_twin) return MCVC_ERR_INVALID;
    TwinCtx* t = g_mcvc_twin;
    g_mcvc_twin = nullptr;
    const int phase = t->phase;
    t->phase = 0;
    if (t->err) return t->err;
    if (phase != 2 || t->next != t->recs.size()) return MCVC_ERR_INVALID;     // the second walk issued fewer launches than the first
    return 0;
}
int mcvc_twin_launches(void) { return (int)t_twin_ctx.recs.size(); }

int mcvc_set_deterministic(int on) { const int was = g_deterministic; g_deterministic = on ? 1 : 0; return was; }
int mcvc_get_deterministic(void) { return g_deterministic; }
int mcvc_set_precise(int on) { const int was = g_precise; g_precise = on ? 1 : 0; return was; }
int mcvc_get_precise(void) { return g_precise; }
int mcvc_set_trunk_persistent(int on) { const int was = g_trunk_net; g_trunk_net = (on == 2) ? 2 : (on ? 1 : 0); return was; }

long long mcvc_gen_packed_floats(void) { return gen_net().packed_floats; }
long long mcvc_disc_packed_floats(void) { return disc_net().packed_floats; }
int mcvc_gen_out_frames(int T) { return gen_dims(1, T).Wu2; }
int mcvc_disc_out_frames(int T) { return disc_dims(1, T).W[3]; }

long long mcvc_gen_stash_floats(int B, int T) { return gen_stash(gen_dims(B, T)).total; }

long long mcvc_gen_scratch_floats(int B, int T)
{
    const Needs nd = gen_needs(B, T);
    return gen_scratch(gen_dims(B, T)).slabs + nd.slab + nd.sg + nd.sgw + nd.wslab + 64;
}

long long mcvc_disc_stash_floats(int B, int T)
{
    const DiscDims d = disc_dims(B, T);
    return disc_stash(d).total + (((long long)B * 80 * T + 3) & ~3LL);   // + a copy of the input for the first layer's weight gradient
}

long long mcvc_disc_scratch_floats(int B, int T)
{
    const Needs nd = disc_needs(B, T);
    return disc_scratch(disc_dims(B, T)).slabs + nd.slab + nd.sg + nd.sgw + nd.wslab + 64;
}

int mcvc_gen_pack(const float* const* params, float* packed, void* stream)
{
    int err = 0;
    const DevPackTable* t = dev_pack_table(0, [](PackTable& pt) {
        const GenNet& g = gen_net();
        const ConvSpec* all[] = {&g.conv1, &g.ds1, &g.ds2, &g.c2d1d, &g.c1d2d, &g.up1, &g.up2, &g.last};
        for (const ConvSpec* c : all) add_spec_jobs(pt, *c);
        for (int i = 0; i < 6; ++i) { add_spec_jobs(pt, g.res_vg[i]); add_spec_jobs(pt, g.res_out[i]); }
    }, &err);
    if (!t) return err;
    set_pack_skips(packed, 0);
    return pack_net(t, params, packed, (hipStream_t)stream);
}

// 1 if every 1-D layer of a generator pass with batch B at T frames takes the fused trunk path (see trunk_fwd / trunk_dgrad)
int mcvc_gen_trunk_fused(int B, int T)
{
    if (B < 1 || T < 1 || !trunk_enabled()) return 0;
    const GenNet& g = gen_net();
    const GenDims d = gen_dims(B, T);
    const int W4 = d.W4;
    bool ok = true;
    for (int i = 0; i < 6 && ok; ++i) {
        ok = ok && mcvc_trunk_applies(g.res_vg[i].Cin, 3, g.res_vg[i].Cout, B, W4, TRUNK_IN_GLU, 1);
        ok = ok && mcvc_trunk_applies(g.res_out[i].Cin, 3, g.res_out[i].Cout, B, W4, TRUNK_IN, 1);
        ok = ok && mcvc_trunk_applies(g.res_out[i].cout_tot, 3, g.res_out[i].Cin, B, W4, TRUNK_PLAIN, 1);
        ok = ok && trunk_pick_ksplit(g.res_vg[i].cout_tot, 3, g.res_vg[i].Cin, B, W4, 0) >= 1;
    }
    ok = ok && mcvc_trunk_applies(g.c1d2d.Cin, 1, g.c1d2d.Cout, B, W4, TRUNK_IN, 1);
    ok = ok && trunk_pick_ksplit(g.c1d2d.cout_tot, 1, g.c1d2d.Cin, B, W4, 0) >= 1;       // its data-gradient (K = 5120)
    ok = ok && trunk_pick_ksplit(g.c2d1d.Cin, 1, g.c2d1d.Cout, B, W4, 1) >= 1;           // conv2dto1d forward (K = 5120)
    ok = ok && mcvc_trunk_applies(g.c2d1d.cout_tot, 1, g.c2d1d.Cin, B, W4, TRUNK_PLAIN, 1);
    return ok ? 1 : 0;
}

// bit 0: the persistent forward trunk kernel would run a (B, T) pass, bit 1: the persistent backward -- under the current switches
// (mcvc_set_trunk_persistent) and the residency bound (mcvc_set_trunk_passes_in_flight x 64 workgroups <= compute units)
int mcvc_gen_trunk_persistent(int B, int T)
{
    if (B < 1 || T < 1 || !trunk_enabled()) return 0;
    const int W4 = gen_dims(B, T).W4;
    int m = 0;
    if (trunk_net_enabled() && mcvc_trunk_net_applies(B, W4)) m |= 1;
    if (trunk_bwd_net_enabled() && mcvc_trunk_bwd_net_applies(B, W4)) m |= 2;
    return m;
}

// `sets`: 1 = only what a forward pass reads, 2 = only what a backward pass reads, 3 = both (see add_spec_jobs).  After a forward-only
// refresh the backward sets are marked stale (bit 2 of the registry) and a backward pass on this buffer fails until sets = 2 has run.
int mcvc_gen_pack_sets(const float* const* params, float* packed, int max_batch, int T, int sets, void* stream)
{
    return mcvc_gen_pack_ranges(params, packed, max_batch, T, sets, 7, stream);
}

// ... restricted to the layers of some parameter ranges (bit 0: upSample1/2 + lastConvLayer = parameters [100,110); bit 1: the residual
// blocks + conv1dto2d = [24,100); bit 2: conv1, downSample1/2, conv2dto1d = [0,24) -- the ranges whose gradients become final one after the
// other during a backward pass, mcvc_gen_backward_overlap): the optimizer step + re-pack of a range can then run beside the rest of the pass
struct GenPackCfg { bool fused, wino_only, w4, w43, up1_w4, up1_w2, up2_w2; int skipped; };
static GenPackCfg gen_pack_cfg(int max_batch, int T)
{
    GenPackCfg q{};
    q.fused = true;
    for (int b = 1; b <= max_batch; ++b)
        if (!mcvc_gen_trunk_fused(b, T)) q.fused = false;
    // every 5x5 layer of every such pass runs on the Winograd kernels (conv_wino / the wino3 branches take no fallback) when
    // the frame count keeps all image sizes even, the tile counts are inside the kernels' range and nothing was switched
    // off through the MCVC_WINO* knobs: then their direct K-major copies are not refreshed either
    static const bool knobs_default = !mcvc_knob_set("MCVC_WINO") && !mcvc_knob_set("MCVC_WINO3") && !mcvc_knob_set("MCVC_WINO3_FWD") && !mcvc_knob_set("MCVC_WINO_GEMM");
    const GenDims dm = gen_dims(max_batch, T);
    const bool wino_default = knobs_default && !g_precise;
    q.wino_only = q.fused && wino_default && (T % 4) == 0 && T >= 32 && (long long)max_batch * 20 * dm.W4 <= 16384;
    // the 64-point weight sets of upSample1/2 only when some pass can take the F(4x4,5x5) path (wino4_applies); otherwise marked absent (bit 16)
    q.w4 = wino_default && wino4_min_nb() > 0 && max_batch >= wino4_min_nb() && (T % 16) == 0;
    q.w43 = wino_default && wino43_min_nb() > 0 && max_batch >= wino43_min_nb() && (T % 16) == 0;      // (bit 32)
    q.up1_w4 = (long long)max_batch * 5 * (T / 16) >= wino4_min_tiles();             // upSample1 runs on 20 x T/4 images: 5 x T/16 tiles per sample
    // the 36-point sets of a layer only where some pass still runs F(2x2,5x5): a layer with >= wino4_min_tiles() tiles PER SAMPLE (upSample2
    // at 64 frames: 80) takes the 4 x 4 scheme in every pass, forward, data gradient and weight gradient (wino4_applies), so its 36-point
    // sets -- 2 x 36/25 of its weights per step -- have no reader (bit 128; r5)
    const bool all4 = q.w4 && wino4_min_nb() == 1;
    q.up1_w2 = !(all4 && 5LL * (T / 16) >= wino4_min_tiles());
    q.up2_w2 = !(all4 && 10LL * (T / 8) >= wino4_min_tiles());
    q.skipped = (q.fused ? (q.wino_only ? 3 : 1) : 0) | (q.w4 ? 0 : 16) | (q.w43 ? 0 : 32) | (q.up1_w2 ? 0 : 128) | (q.up2_w2 ? 0 : 256);
    return q;
}
// range_mask bits: 1 = parameters [100,110), 2 = [24,100), 4 = [0,24); r4: the head in three parts, in the order their gradients become final
// during a backward pass (MCVC_BWD_FINE_MILESTONES): 8 = [12,24) (downSample2, conv2dto1d), 16 = [4,12) (downSample1), 32 = [0,4) (conv1)
static bool gen_range_mask_ok(int m) { return m >= 1 && m <= 63 && !((m & 4) && (m & 56)); }
static bool gen_in_range(int m, int p)
{
    return ((m & 1) && p >= 100 && p < 110) || ((m & 2) && p >= 24 && p < 100) || ((m & 4) && p < 24) ||
           ((m & 8) && p >= 12 && p < 24) || ((m & 16) && p >= 4 && p < 12) || ((m & 32) && p < 4);
}
static void gen_pack_build(PackTable& pt, const GenPackCfg& q, int sets, int range_mask)
{
    const GenNet& g = gen_net();
    if (range_mask & (4 | 32)) add_spec_jobs(pt, g.conv1, false, q.wino_only, sets, q.w4, q.w43);
    if (range_mask & (4 | 16)) add_spec_jobs(pt, g.ds1, false, q.wino_only, sets, q.w4, q.w43);
    if (range_mask & (4 | 8)) {
        add_spec_jobs(pt, g.ds2, false, q.wino_only, sets, q.w4, q.w43);
        add_spec_jobs(pt, g.c2d1d, q.fused, false, sets);
    }
    if (range_mask & 1) {
        const ConvSpec* up[] = {&g.up1, &g.up2, &g.last};
        // (a layer whose largest pass has fewer than 64 F(4x4) tiles never takes that path, conv_wino4: its 64-point sets are not written)
        for (const ConvSpec* c : up) add_spec_jobs(pt, *c, false, q.wino_only, sets, q.w4 && (c != &g.up1 || q.up1_w4), q.w43, false,
                                                   c == &g.up1 ? q.up1_w2 : (c == &g.up2 ? q.up2_w2 : true));
    }
    if (range_mask & 2) {
        add_spec_jobs(pt, g.c1d2d, q.fused, false, sets);
        for (int i = 0; i < 6; ++i) { add_spec_jobs(pt, g.res_vg[i], q.fused, false, sets); add_spec_jobs(pt, g.res_out[i], q.fused, false, sets); }
    }
}
static int gen_pack_key(const GenPackCfg& q, int sets, int range_mask)
{
    return 16 + 4 * sets + (q.fused ? (q.wino_only ? 3 : 2) : 0) + 64 * range_mask /* <= 63 */ + (q.w4 ? 4096 : 0) + (q.w43 ? 8192 : 0) + (q.up1_w4 ? 16384 : 0) +
           (q.up1_w2 ? 0 : 32768) + (q.up2_w2 ? 0 : 65536);
}

int mcvc_gen_pack_ranges(const float* const* params, float* packed, int max_batch, int T, int sets, int range_mask, void* stream)
{
    if (sets < 1 || sets > 3 || !gen_range_mask_ok(range_mask)) return MCVC_ERR_INVALID;
    const GenPackCfg q = gen_pack_cfg(max_batch, T);
    int err = 0;
    const DevPackTable* t = dev_pack_table(gen_pack_key(q, sets, range_mask), [&](PackTable& pt) { gen_pack_build(pt, q, sets, range_mask); }, &err);
    if (!t) return err;
    if (sets == 1) set_pack_skips(packed, q.skipped | 4);
    else if (sets == 2) set_pack_skips(packed, get_pack_skips(packed) & ~4);
    else set_pack_skips(packed, q.skipped);
    if (t->njobs == 0) return 0;
    return pack_net(t, params, packed, (hipStream_t)stream);
}

// optimizer.step() of a generator's parameter ranges (train.py:242) fused with the refresh of every packed copy derived from them: ONE
// launch, one reader and one writer per weight (update_net_kernel).  `numel`: element count of each of the 110 parameter tensors;
// flat / grad / grad2 (nullable) / exp_avg / exp_avg_sq: the flat buffers the parameters, their gradients and moments live in at EQUAL
// offsets (the engine's layout): the gradient of params[i] is grad + (params[i] - flat).  zero_grads: clear the gradient(s) behind the read.
int mcvc_gen_update_ranges(const float* const* params, const long long* numel, float* packed, int max_batch, int T, int range_mask,
                           const float* flat, float* grad, float* grad2, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                           float eps, int step, float grad_scale, int zero_grads, void* stream)
{
    if (!params || !numel || !packed || !gen_range_mask_ok(range_mask)) return MCVC_ERR_INVALID;
    const GenPackCfg q = gen_pack_cfg(max_batch, T);
    int err = MCVC_ERR_INVALID;
    auto in_range = [range_mask](int p) { return gen_in_range(range_mask, p); };
    const DevUpdTable* t = dev_upd_table(gen_pack_key(q, 3, range_mask), [&](PackTable& pt) { gen_pack_build(pt, q, 3, range_mask); }, numel, MCVC_GEN_NPARAMS,
                                         in_range, &err);
    if (!t) return err;
    // (a range's update leaves ITS copies fresh; the registry word describes which KINDS of copies exist, as after a full re-pack)
    set_pack_skips(packed, q.skipped);
    return update_net(t, params, numel, packed, UpdOpt{flat, grad, grad2, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, zero_grads}, (hipStream_t)stream);
}

int mcvc_gen_pack_small_batch(const float* const* params, float* packed, int max_batch, int T, void* stream)
{
    return mcvc_gen_pack_sets(params, packed, max_batch, T, 3, stream);
}

int mcvc_disc_pack(const float* const* params, float* packed, void* stream)
{
    int err = 0;
    const DevPackTable* t = dev_pack_table(1, [](PackTable& pt) {
        const DiscNet& n = disc_net();
        add_spec_jobs(pt, n.conv1);
        for (int i = 0; i < 3; ++i) add_spec_jobs(pt, n.ds[i]);
        add_spec_jobs(pt, n.outc);
    }, &err);
    if (!t) return err;
    set_pack_skips(packed, 0);
    return pack_net(t, params, packed, (hipStream_t)stream);
}

// Discriminator re-pack restricted to what the implicit-GEMM schedule reads: of the three stride-2 layers (99.9 % of the weights) only the
// tap-major FORWARD copy + bias -- it serves the forward pass and, read row-major, the data gradient (sgemm.h arow); the weight gradient reads
// no weights at all; the first / output layers (tiny) are refreshed in full.  Falls back to the full pack when a layer would not take that
// path at T frames.  The direct kernels' data-gradient copies are then marked stale (bit 8) and refuse to run.
int mcvc_disc_pack_small(const float* const* params, float* packed, int T, void* stream) { return mcvc_disc_pack_batch(params, packed, 1, T, stream); }

// which job table a discriminator re-pack for passes of up to max_batch samples uses: 1 = full, 4 = implicit GEMMs; `skips` = the registry
// word it leaves
static int disc_pack_kind(int max_batch, int T, int* skips)
{
    const DiscDims d = disc_dims(max_batch < 1 ? 1 : max_batch, T);
    if (disc_igemm(d)) { *skips = 8; return 4; }
    *skips = 0;
    return 1;
}
static void disc_pack_build(PackTable& pt, int kind)
{
    const DiscNet& n = disc_net();
    add_spec_jobs(pt, n.conv1);
    for (int i = 0; i < 3; ++i) {
        if (kind == 1) add_spec_jobs(pt, n.ds[i]);
        else add_spec_jobs(pt, n.ds[i], false, false, 1, true, true, true);
    }
    add_spec_jobs(pt, n.outc);
}

int mcvc_disc_pack_batch(const float* const* params, float* packed, int max_batch, int T, void* stream)
{
    int skips = 0, err = 0;
    const int kind = disc_pack_kind(max_batch, T, &skips);
    if (kind == 1) { set_pack_skips(packed, 0); return mcvc_disc_pack(params, packed, stream); }
    const DevPackTable* t = dev_pack_table(kind, [kind](PackTable& pt) { disc_pack_build(pt, kind); }, &err);
    if (!t) return err;
    set_pack_skips(packed, skips);
    return pack_net(t, params, packed, (hipStream_t)stream);
}

// optimizer.step() of one discriminator (train.py:299) fused with the refresh of its packed copies; arguments as mcvc_gen_update_ranges
// (`numel`: 20 entries, 0 for the parameters of the unused downSample4 block, which are neither updated nor packed).
int mcvc_disc_update_batch(const float* const* params, const long long* numel, float* packed, int max_batch, int T,
                           const float* flat, float* grad, float* grad2, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                           float eps, int step, float grad_scale, int zero_grads, void* stream)
{
    if (!params || !numel || !packed) return MCVC_ERR_INVALID;
    int skips = 0, err = MCVC_ERR_INVALID;
    const int kind = disc_pack_kind(max_batch, T, &skips);
    const DevUpdTable* t = dev_upd_table(kind, [kind](PackTable& pt) { disc_pack_build(pt, kind); }, numel, MCVC_DISC_NPARAMS, [](int) { return true; }, &err);
    if (!t) return err;
    set_pack_skips(packed, skips);
    return update_net(t, params, numel, packed, UpdOpt{flat, grad, grad2, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, zero_grads}, (hipStream_t)stream);
}

int mcvc_gen_forward(const float* const* params, const float* packed, const float* x, const float* mask, float* out, float* stash,
                     float* scratch, long long scratch_floats, int B, int T, void* stream)
{
    if (B < 1 || T < 1 || !params || !packed || !x || !out || !stash || !scratch) return MCVC_ERR_INVALID;
    const GenDims d = gen_dims(B, T);
    Exec ex = make_exec(stream, nullptr, scratch, scratch_floats, gen_scratch(d).slabs, gen_needs(B, T));
    if (ex.wslab_cap < 0) return MCVC_ERR_WORKSPACE;
    { const GenScratch q = gen_scratch(d); ex.wv = scratch + q.wv; ex.wm = scratch + q.wm; ex.wino_cap = q.wino_floats;
      ex.sync = reinterpret_cast<unsigned*>(scratch + q.sync); }
    ex.pack_skips = get_pack_skips(packed);
    gen_forward_impl(ex, params, packed, x, mask, out, stash, d);
    return ex.err;
}

// Error word of the persistent trunk kernels in a generator scratch buffer (sticky until reset): 0 = no fault, 1 + l = a workgroup gave
// up waiting for the arrivals of layer l (the pass's output was poisoned with NaN).  SYNCHRONOUS (a 4-byte device-to-host copy on `stream`).
int mcvc_gen_trunk_fault(float* scratch, int B, int T, int reset, void* stream)
{
    if (!scratch || B < 1 || T < 1) return -MCVC_ERR_INVALID;
    unsigned* w = reinterpret_cast<unsigned*>(scratch + gen_scratch(gen_dims(B, T)).sync) + MCVC_TRUNK_SYNC_WORDS - 1;
    unsigned v = 0;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(&v, w, sizeof(v), hipMemcpyDeviceToHost, s) != hipSuccess) return -MCVC_ERR_INVALID;
    if (reset && hipMemsetAsync(w, 0, sizeof(v), s) != hipSuccess) return -MCVC_ERR_INVALID;
    if (hipStreamSynchronize(s) != hipSuccess) return -MCVC_ERR_INVALID;
    return (int)(v & 0x7fffffffu);             // (an un-initialised word -- first read after allocation -- must not look like an error return)
}
int mcvc_debug_trunk_fault_inject(int on) { return mcvc_trunk_set_fault_inject(on); }
int mcvc_set_trunk_passes_in_flight(int n) { return mcvc_trunk_set_passes_in_flight(n); }

int mcvc_gen_backward_overlap(const float* const* params, const float* packed, float* const* grads, const float* mask, const float* dout,
                              float* dx, int accumulate_dx, const float* stash, float* scratch, long long scratch_floats, int B, int T,
                              void* stream, void* aux_stream, void* const* milestones)
{
    return mcvc_gen_backward_flags(params, packed, grads, mask, dout, dx, accumulate_dx, stash, scratch, scratch_floats, B, T, stream, aux_stream,
                                   milestones, 0);
}

int mcvc_gen_backward_flags(const float* const* params, const float* packed, float* const* grads, const float* mask, const float* dout,
                            float* dx, int accumulate_dx, const float* stash, float* scratch, long long scratch_floats, int B, int T,
                            void* stream, void* aux_stream, void* const* milestones, int flags)
{
    return mcvc_gen_backward_window(params, packed, grads, mask, dout, dx, accumulate_dx, stash, B, 0, scratch, scratch_floats, B, T, stream, aux_stream,
                                    milestones, flags);
}

int mcvc_gen_backward_prefix(const float* const* params, const float* packed, float* const* grads, const float* mask, const float* dout,
                             float* dx, int accumulate_dx, const float* stash, int stash_B, float* scratch, long long scratch_floats, int B, int T,
                             void* stream, void* aux_stream, void* const* milestones, int flags)
{
    return mcvc_gen_backward_window(params, packed, grads, mask, dout, dx, accumulate_dx, stash, stash_B, 0, scratch, scratch_floats, B, T, stream,
                                    aux_stream, milestones, flags);
}

int mcvc_gen_backward_window(const float* const* params, const float* packed, float* const* grads, const float* mask, const float* dout,
                             float* dx, int accumulate_dx, const float* stash, int stash_B, int stash_b0, float* scratch, long long scratch_floats,
                             int B, int T, void* stream, void* aux_stream, void* const* milestones, int flags)
{
    if (B < 1 || T < 1 || stash_b0 < 0 || stash_B < stash_b0 + B || !params || !packed || !dout || !stash || !scratch) return MCVC_ERR_INVALID;
    const GenDims d = gen_dims(B, T);
    Exec ex = make_exec(stream, aux_stream, scratch, scratch_floats, gen_scratch(d).slabs, gen_needs(B, T));
    if (ex.wslab_cap < 0) return MCVC_ERR_WORKSPACE;
    { const GenScratch q = gen_scratch(d); ex.wv = scratch + q.wv; ex.wm = scratch + q.wm; ex.wino_cap = q.wino_floats;
      ex.wv2 = scratch + q.wv2; ex.wm2 = scratch + q.wm2; ex.wu = scratch + q.wu; ex.wu_cap = q.wu_floats;
      ex.sync = reinterpret_cast<unsigned*>(scratch + q.sync); }
    ex.pack_skips = get_pack_skips(packed);
    if (ex.pack_skips & 4) return MCVC_ERR_INVALID;          // forward-only re-pack: the backward sets are stale (mcvc_gen_pack_sets)
    ex.no_join = (flags & 1) && aux_stream;
    ex.fine_ms = (flags & 2) != 0;
    gen_backward_impl(ex, params, packed, grads, mask, dout, dx, accumulate_dx, stash, scratch, d, milestones, stash_B, stash_b0);
    return ex.err;
}

int mcvc_gen_backward(const float* const* params, const float* packed, float* const* grads, const float* mask, const float* dout,
                      float* dx, int accumulate_dx, const float* stash, float* scratch, long long scratch_floats, int B, int T, void* stream,
                      void* aux_stream)
{
    return mcvc_gen_backward_overlap(params, packed, grads, mask, dout, dx, accumulate_dx, stash, scratch, scratch_floats, B, T, stream,
                                     aux_stream, nullptr);
}

int mcvc_disc_forward(const float* const* params, const float* packed, const float* x, float* out, float* stash, float* scratch,
                      long long scratch_floats, int B, int T, void* stream)
{
    if (B < 1 || T < 1 || !params || !packed || !x || !out || !stash || !scratch) return MCVC_ERR_INVALID;
    const DiscDims d = disc_dims(B, T);
    Exec ex = make_exec(stream, nullptr, scratch, scratch_floats, disc_scratch(d).slabs, disc_needs(B, T));
    if (ex.wslab_cap < 0) return MCVC_ERR_WORKSPACE;
    ex.pack_skips = get_pack_skips(packed);
    // keep the input for the first layer's weight gradient
    ex.fail(mcvc_copy_launch(x, stash + disc_stash(d).total, B * 80 * T, ex.s));
    disc_forward_impl(ex, params, packed, x, out, stash, d);
    return ex.err;
}

int mcvc_disc_backward(const float* const* params, const float* packed, float* const* grads, const float* dout, int dout_is_logit_grad,
                       float* dx, int accumulate_dx, const float* stash, float* scratch, long long scratch_floats, int B, int T, void* stream,
                       void* aux_stream)
{
    if (B < 1 || T < 1 || !params || !packed || !dout || !stash || !scratch) return MCVC_ERR_INVALID;
    const DiscDims d = disc_dims(B, T);
    Exec ex = make_exec(stream, aux_stream, scratch, scratch_floats, disc_scratch(d).slabs, disc_needs(B, T));
    if (ex.wslab_cap < 0) return MCVC_ERR_WORKSPACE;
    ex.pack_skips = get_pack_skips(packed);
    disc_backward_impl(ex, params, packed, grads, dout, dout_is_logit_grad, dx, accumulate_dx, stash, scratch, d);
    return ex.err;
}

int mcvc_l1_loss(const float* a, const float* b, long long n, float weight, float* loss_slot, float* term_slot, float* grad_a,
                 int accumulate_grad, void* stream)
{
    return mcvc_l1_loss_launch(a, b, n, weight, loss_slot, term_slot, grad_a, accumulate_grad, (hipStream_t)stream);
}

int mcvc_loss_combine(const float* pairs, int n, const int* loss_dst, const int* term_dst, float* slots, void* stream)
{
    if (!pairs || !slots || !loss_dst || !term_dst) return MCVC_ERR_INVALID;
    return mcvc_loss_combine_launch(pairs, n, loss_dst, term_dst, slots, (hipStream_t)stream);
}

int mcvc_lsgan_loss(const float* d, long long n, float target, float weight, float* loss_slot, float* term_slot, float* grad_logit, void* stream)
{
    return mcvc_lsgan_loss_launch(d, n, target, weight, loss_slot, term_slot, grad_logit, (hipStream_t)stream);
}

int mcvc_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale, void* stream)
{
    return mcvc_adam_launch(p, const_cast<float*>(g), nullptr, 0, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)stream);
}

int mcvc_adam_step2(float* p, float* g, float* g2, int zero_grads, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                    float eps, int step, float grad_scale, void* stream)
{
    return mcvc_adam_launch(p, g, g2, zero_grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)stream);
}

int mcvc_draw_batch(const float* bank_A, const int* offs_A, int n_A, long long frames_A, const float* bank_B, const int* offs_B, int n_B,
                    long long frames_B, int B, int T, int max_mask_len, unsigned long long seed, unsigned long long step,
                    float* real_A, float* mask_A, float* real_B, float* mask_B, int* draws, void* stream)
{
    if (!bank_A || !offs_A || !bank_B || !offs_B || n_A < 1 || n_B < 1 || !real_A || !mask_A || !real_B || !mask_B) return MCVC_ERR_INVALID;
    DrawArgs a{};
    a.bank[0] = bank_A; a.bank[1] = bank_B; a.offs[0] = offs_A; a.offs[1] = offs_B; a.ld[0] = frames_A; a.ld[1] = frames_B;
    a.n[0] = n_A; a.n[1] = n_B; a.B = B; a.T = T; a.max_mask_len = max_mask_len; a.seed = seed; a.step = step;
    a.real[0] = real_A; a.real[1] = real_B; a.mask[0] = mask_A; a.mask[1] = mask_B; a.draws = draws;
    return mcvc_draw_batch_launch(a, (hipStream_t)stream);
}

int mcvc_axpy(float* y, const float* x, float alpha, long long n, void* stream)
{
    return mcvc_axpy_launch(y, x, alpha, n, (hipStream_t)stream);
}

// ---- single-op entry points ----------------------------------------------------------------------------
static ConvSpec single_spec(int Cout, int Cin, int KH, int KW, int stride, int ph, int pw)
{
    ConvSpec c = mk(Cin, Cout, 1, KH, KW, stride, ph, pw, 0, 1, -1, -1, 1);
    long long cur = 0;
    spec_finalize(c, cur);
    c.off_dcls = -1;               // (the single-op pack holds the merged matrix only)
    return c;
}

long long mcvc_conv2d_pack_floats(int Cout, int Cin, int KH, int KW)
{
    // stride-2 classes partition the taps (same rows in total) but each class carries its own zero pad row
    ConvSpec c = mk(Cin, Cout, 1, KH, KW, 1, 0, 0, 0, 1, -1, -1, 1);
    long long cur = 0;
    spec_finalize(c, cur);
    // stride 2 uses the merged parity layout: at most (KH+1)/2+1 x (KW+1)/2+1 taps x 4*Cin columns
    const long long merged = ((long long)c.dg_rows_co * ((KH + 1) / 2 + 1) * ((KW + 1) / 2 + 1) + 1) * round_up_i(4 * Cin, 32);
    return cur + merged + 4LL * c.cin_pk + 64;
}

int mcvc_conv2d_forward(const float* x, const float* w, const float* bias, float* y, float* wpack, float* slabs, int max_slabs,
                        int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int pixel_shuffle, void* stream)
{
    if (stride != 1 && stride != 2) return MCVC_ERR_INVALID;
    const ConvSpec c = single_spec(Cout, Cin, KH, KW, stride, pad_h, pad_w);
    Exec ex{}; ex.s = (hipStream_t)stream;
    ex.fail(mcvc_pack_fwd_launch(w, wpack + c.off_fwd, Cout, Cin * KH * KW, c.cout_pk, 0, ex.s));
    if (bias) ex.fail(mcvc_copy_launch(bias, wpack + c.off_bias, Cout, ex.s));
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    const long long y_total = (long long)N * Cout * OH * OW;
    ex.slabs = slabs; ex.slab_cap = slabs ? (long long)(max_slabs - 1) * y_total : 0; ex.max_split = max_slabs > 0 ? max_slabs : 1;
    int ns = 1;
    View yv = pixel_shuffle ? View{y, (long long)Cout * OH * OW, 4LL * OH * OW, 2 * OW} : View{y, (long long)Cout * OH * OW, (long long)OH * OW, OW};
    ConvProblem p{Cin, H, W, Cout, OH, OW, KH, KW, stride, pad_h, pad_w};
    ConvIO io{};
    io.x = x; io.x_sb = (long long)Cin * H * W; io.x_sc = (long long)H * W; io.x_sh = W;
    io.y = yv.p; io.y_sb = yv.sb; io.y_sc = yv.sc; io.y_sh = yv.sh; io.y_sw = 1; io.shuffle = pixel_shuffle;
    int want = (slabs && max_slabs > 1) ? mcvc_conv_plan_nsplit(p, N, 1) : 1;
    if (want > max_slabs) want = max_slabs;
    run_conv(ex, p, N, io, y_total, wpack + c.off_fwd, c.w_rows, c.cout_pk, bias ? wpack + c.off_bias : nullptr, 0, want, &ns);
    if (ns > 1) act_fwd(ex, y, y_total, ns, nullptr, 1, 1, (int)y_total, ACT_NONE);
    return ex.err;
}

int mcvc_conv2d_dgrad(const float* dy, const float* w, float* dx, float* wpack, float* slabs, int max_slabs, int N, int Cin, int H, int W,
                      int Cout, int KH, int KW, int stride, int pad_h, int pad_w, void* stream)
{
    if (stride != 1 && stride != 2) return MCVC_ERR_INVALID;
    const ConvSpec c = single_spec(Cout, Cin, KH, KW, stride, pad_h, pad_w);
    Exec ex{}; ex.s = (hipStream_t)stream;
    PackDgradArgs a{};
    a.Cin = Cin; a.KH = KH; a.KW = KW; a.step = stride; a.ld = c.merged ? c.mg_ld : c.cin_pk; a.co_off = 0; a.ncls = c.ncls;
    a.merged = c.merged; a.mg_kh = c.mg_kh; a.mg_kw = c.mg_kw;
    for (int k = 0; k < c.ncls; ++k) a.cls[k] = c.cls[k];
    ex.fail(mcvc_pack_dgrad_launch(w, wpack + c.off_dgrad, a, Cout, ex.s));
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    const long long dx_total = (long long)N * Cin * H * W;
    ex.slabs = slabs; ex.slab_cap = slabs ? (long long)(max_slabs - 1) * dx_total : 0; ex.max_split = max_slabs > 0 ? max_slabs : 1;
    int ns = 1;
    // positions of dx never written by any parity class do not exist: classes tile the whole input grid
    conv_dgrad(ex, c, wpack, N, H, W, CView{dy, (long long)Cout * OH * OW, (long long)OH * OW, OW},
               View{dx, (long long)Cin * H * W, (long long)H * W, W}, dx_total, 0, (slabs && max_slabs > 1) ? 1 : 0, &ns);
    if (ns > max_slabs && !ex.err) return MCVC_ERR_WORKSPACE;
    if (ns > 1) act_fwd(ex, dx, dx_total, ns, nullptr, 1, 1, (int)dx_total, ACT_NONE);
    return ex.err;
}

long long mcvc_conv2d_wgrad_slab_floats(int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_h, int pad_w)
{
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    ConvProblem p{Cin, H, W, Cout, OH, OW, KH, KW, stride, pad_h, pad_w};
    return mcvc_wgrad_plan_slab_floats(p, N);
}

int mcvc_conv2d_wgrad(const float* x, const float* dy, float* dw, float* slabs, long long slab_floats, int N, int Cin, int H, int W, int Cout,
                      int KH, int KW, int stride, int pad_h, int pad_w, void* stream)
{
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    ConvProblem p{Cin, H, W, Cout, OH, OW, KH, KW, stride, pad_h, pad_w};
    WgradIO io{x, (long long)Cin * H * W, (long long)H * W, W, dy, (long long)Cout * OH * OW, (long long)OH * OW, OW};
    // (the same special cases the network planner takes, conv_wgrad: dw accumulates, the caller passes zeros)
    if (Cout == 1 && mcvc_wgrad_cout1_applies(p)) return mcvc_wgrad_cout1_launch(p, N, io, dw, (hipStream_t)stream);
    if (mcvc_wgrad_cin2_applies(p, io)) return mcvc_wgrad_cin2_launch(p, N, io, dw, (hipStream_t)stream);
    return mcvc_wgrad_launch(p, N, io, dw, slabs, slab_floats, (hipStream_t)stream);
}

// ---- one convolution LAYER of the path, op by op, through the planner the networks use (conv_fwd / conv_dgrad / conv_wgrad) ----------
// SURVEY.md section 8(b) names single-op entries for the kernels that are hot at the trainer's shapes; the Winograd and staged-GEMM forms
// live inside the planner (their weight sets are job kinds of the whole-network pack), so the op-level ABI is the planner on ONE layer.
namespace {
struct LayerCtx { ConvSpec c; long long packed_floats; };
static LayerCtx layer_ctx(int Cin, int Cout, int nbr, int KH, int KW, int stride, int ph, int pw)
{
    LayerCtx l{};
    l.c = mk(Cin, Cout, nbr, KH, KW, stride, ph, pw, 0, 1, nbr == 2 ? 2 : -1, nbr == 2 ? 3 : -1, 1);
    long long cur = 0;
    spec_finalize(l.c, cur);
    l.packed_floats = cur + 64;
    return l;
}
struct LayerScratch { long long wv, wm, wv2, wm2, wino_floats, wu, wu_floats, tail; };
static LayerScratch layer_scratch(const ConvSpec& c, int N, int H, int W)
{
    LayerScratch q{};
    long long cur = 0;
    auto take = [&](long long n) { const long long o = cur; cur += (n + 3) & ~3LL; return o; };
    const long long tiles2 = (((long long)N * ((H + 1) / 2) * ((W + 1) / 2) + 63) & ~63LL) + 64;
    const long long chan = std::max<long long>(std::max(4LL * c.Cin, (long long)c.cout_tot), 64);
    q.wino_floats = 36LL * chan * tiles2;
    q.wv = take(q.wino_floats); q.wm = take(q.wino_floats); q.wv2 = take(q.wino_floats); q.wm2 = take(q.wino_floats);
    q.wu_floats = 64LL * c.cout_tot * 4 * c.Cin;
    q.wu = take(q.wu_floats);
    q.tail = cur;
    return q;
}
static Needs layer_needs(const ConvSpec& c, int N, int H, int W, int scheme)
{
    Exec ex{}; ex.dry = true; ex.force_scheme = scheme;
    const LayerScratch q = layer_scratch(c, N, H, W);
    ex.wv = ex.wm = ex.wv2 = ex.wm2 = ex.wu = reinterpret_cast<float*>(16);      // (non-null: the dry run only looks at capacities)
    ex.wino_cap = q.wino_floats; ex.wu_cap = q.wu_floats;
    const int OH = conv_out(H, c.KH, c.stride, c.ph), OW = conv_out(W, c.KW, c.stride, c.pw);
    int ns = 1;
    conv_fwd(ex, c, nullptr, N, H, W, CView{nullptr, (long long)c.Cin * H * W, (long long)H * W, W},
             View{nullptr, (long long)c.cout_tot * OH * OW, (long long)OH * OW, OW}, (long long)N * c.cout_tot * OH * OW, 0, 1, &ns);
    conv_dgrad(ex, c, nullptr, N, H, W, CView{nullptr, (long long)c.cout_tot * OH * OW, (long long)OH * OW, OW},
               View{nullptr, (long long)c.Cin * H * W, (long long)H * W, W}, (long long)N * c.Cin * H * W, 0, 1, &ns);
    conv_wgrad(ex, c, nullptr, N, H, W, CView{nullptr, (long long)c.Cin * H * W, (long long)H * W, W},
               CView{nullptr, (long long)c.cout_tot * OH * OW, (long long)OH * OW, OW});
    if (igemm_applies(c, H, W)) {
        conv_fwd_igemm(ex, c, nullptr, N, H, W, nullptr, View{nullptr, (long long)c.cout_tot * OH * OW, (long long)OH * OW, OW}, (long long)N * c.cout_tot * OH * OW, 1, &ns);
        conv_dgrad_igemm(ex, c, nullptr, N, H, W, nullptr, View{nullptr, (long long)c.Cin * H * W, (long long)H * W, W}, (long long)N * c.Cin * H * W, 0, 1, &ns);
        ex.wgrad_x_xs = 1;             // (the implicit weight gradient's K-split slabs)
        conv_wgrad(ex, c, nullptr, N, H, W, CView{nullptr, (long long)c.Cin * 4 * mcvc_xs_plane(H, W), 4 * mcvc_xs_plane(H, W), W},
                   CView{nullptr, (long long)c.cout_tot * OH * OW, (long long)OH * OW, OW});
        ex.wgrad_x_xs = 0;
    }
    return Needs{(ex.slab_need + 3) & ~3LL, (ex.wslab_need + 3) & ~3LL, (ex.sg_need + 3) & ~3LL, (ex.sgw_need + 3) & ~3LL};
}
struct SchemeGuard {            // scheme 3: no Winograd; 4: neither Winograd nor staged GEMM (thread-local planner switches)
    int w, g, u;
    explicit SchemeGuard(int scheme) : w(t_no_wino), g(t_no_sgemm), u(t_user_operands) { if (scheme >= 3) t_no_wino = 1; if (scheme == 4) t_no_sgemm = 1; t_user_operands = 1; }
    ~SchemeGuard() { t_no_wino = w; t_no_sgemm = g; t_user_operands = u; }
};
static Exec layer_exec(const ConvSpec& c, int N, int H, int W, int scheme, float* scratch, long long scratch_floats, void* stream, int* err)
{
    const LayerScratch q = layer_scratch(c, N, H, W);
    const Needs nd = layer_needs(c, N, H, W, scheme);
    Exec ex = make_exec(stream, nullptr, scratch, scratch_floats, q.tail, nd);
    if (ex.wslab_cap < 0) { *err = MCVC_ERR_WORKSPACE; return ex; }
    ex.wv = scratch + q.wv; ex.wm = scratch + q.wm; ex.wv2 = scratch + q.wv2; ex.wm2 = scratch + q.wm2; ex.wino_cap = q.wino_floats;
    ex.wu = scratch + q.wu; ex.wu_cap = q.wu_floats;
    ex.force_scheme = scheme;
    return ex;
}
}  // namespace

long long mcvc_layer_packed_floats(int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w)
{
    if (branches < 1 || branches > 2 || (stride != 1 && stride != 2)) return -1;
    return layer_ctx(Cin, Cout, branches, KH, KW, stride, pad_h, pad_w).packed_floats;
}

long long mcvc_layer_scratch_floats(int N, int H, int W, int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w)
{
    if (branches < 1 || branches > 2 || (stride != 1 && stride != 2) || N < 1) return -1;
    const LayerCtx l = layer_ctx(Cin, Cout, branches, KH, KW, stride, pad_h, pad_w);
    long long worst = 0;
    for (int scheme = 0; scheme <= 4; ++scheme) {
        SchemeGuard sg(scheme);
        const Needs nd = layer_needs(l.c, N, H, W, scheme);
        const long long need = nd.slab + nd.wslab + nd.sg + nd.sgw;
        if (need > worst) worst = need;
    }
    ConvProblem p{Cin, H, W, Cout, conv_out(H, KH, stride, pad_h), conv_out(W, KW, stride, pad_w), KH, KW, stride, pad_h, pad_w};
    return layer_scratch(l.c, N, H, W).tail + worst + mcvc_wgrad_plan_slab_floats(p, N) + 1024;
}

int mcvc_layer_pack(const float* w0, const float* b0, const float* w1, const float* b1, float* packed, int Cin, int Cout, int branches,
                    int KH, int KW, int stride, int pad_h, int pad_w, void* stream)
{
    if (!w0 || !b0 || !packed || branches < 1 || branches > 2 || (branches == 2 && (!w1 || !b1))) return MCVC_ERR_INVALID;
    const LayerCtx l = layer_ctx(Cin, Cout, branches, KH, KW, stride, pad_h, pad_w);
    // the job table of this layer alone (cached per device and layer shape like the networks' tables)
    const long long h = (((((((long long)Cin * 31 + Cout) * 31 + branches) * 31 + KH) * 31 + KW) * 31 + stride) * 31 + pad_h) * 31 + pad_w;
    const int key = 1000 + (int)(h % 1000003);
    int err = 0;
    const DevPackTable* t = dev_pack_table(key, [&](PackTable& pt) { add_spec_jobs(pt, l.c); }, &err);
    if (!t) return err ? err : MCVC_ERR_INVALID;
    const float* params[4] = {w0, b0, w1, b1};
    return pack_net(t, params, packed, (hipStream_t)stream);
}

int mcvc_layer_forward(const float* x, const float* packed, const float* w0, const float* w1, float* y, float* scratch, long long scratch_floats,
                       int N, int H, int W, int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w, int scheme,
                       int pixel_shuffle, void* stream)
{
    if (!x || !packed || !y || !scratch || scheme < 0 || scheme > 5 || branches < 1 || branches > 2 || (stride != 1 && stride != 2)) return MCVC_ERR_INVALID;
    const LayerCtx l = layer_ctx(Cin, Cout, branches, KH, KW, stride, pad_h, pad_w);
    if ((scheme == 1 || scheme == 2) && !(l.c.wino || l.c.wino3)) return MCVC_ERR_INVALID;       // no Winograd form of this layer shape
    if (scheme == 5) {          // implicit GEMM (sgemm.h): the dense input is converted to the phase-split padded layout the networks' producers write
        if (!igemm_applies(l.c, H, W) || pixel_shuffle) return MCVC_ERR_INVALID;
        int err5 = 0;
        Exec ex5 = layer_exec(l.c, N, H, W, 0, scratch, scratch_floats, stream, &err5);
        if (err5) return err5;
        if ((long long)N * mcvc_xs_floats(Cin, H, W) > ex5.wino_cap) return MCVC_ERR_WORKSPACE;
        ex5.fail(mcvc_xs_from_dense_launch(x, ex5.wv, N, Cin, H, W, ex5.s));
        const int OH5 = H / 2, OW5 = W / 2;
        const long long yt = (long long)N * l.c.cout_tot * OH5 * OW5;
        int ns5 = 1;
        conv_fwd_igemm(ex5, l.c, packed, N, H, W, ex5.wv, View{y, (long long)l.c.cout_tot * OH5 * OW5, (long long)OH5 * OW5, OW5}, yt, 1, &ns5);
        if (ns5 > 1) act_fwd(ex5, y, yt, ns5, nullptr, 1, 1, (int)yt, ACT_NONE);
        return ex5.err;
    }
    SchemeGuard sg(scheme);
    int err = 0;
    Exec ex = layer_exec(l.c, N, H, W, scheme, scratch, scratch_floats, stream, &err);
    if (err) return err;
    const float* params[4] = {w0, nullptr, w1, nullptr};
    ex.params = params;
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    const long long y_total = (long long)N * l.c.cout_tot * OH * OW;
    View yv = pixel_shuffle ? View{y, (long long)l.c.cout_tot * OH * OW, 4LL * OH * OW, 2 * OW} : View{y, (long long)l.c.cout_tot * OH * OW, (long long)OH * OW, OW};
    int ns = 1;
    conv_fwd(ex, l.c, packed, N, H, W, CView{x, (long long)Cin * H * W, (long long)H * W, W}, yv, y_total, pixel_shuffle, 1, &ns);
    if (ns > 1) act_fwd(ex, y, y_total, ns, nullptr, 1, 1, (int)y_total, ACT_NONE);
    return ex.err;
}

int mcvc_layer_dgrad(const float* dy, const float* packed, const float* w0, const float* w1, float* dx, float* scratch, long long scratch_floats,
                     int N, int H, int W, int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w, int scheme, void* stream)
{
    if (!dy || !packed || !dx || !scratch || scheme < 0 || scheme > 5 || branches < 1 || branches > 2 || (stride != 1 && stride != 2)) return MCVC_ERR_INVALID;
    const LayerCtx l = layer_ctx(Cin, Cout, branches, KH, KW, stride, pad_h, pad_w);
    if (scheme == 5) {          // implicit GEMM: the four parity classes scattered straight into dx; the dense dY is padded first (the networks' norm_bwd writes it padded)
        if (!igemm_applies(l.c, H, W)) return MCVC_ERR_INVALID;
        int err5 = 0;
        Exec ex5 = layer_exec(l.c, N, H, W, 0, scratch, scratch_floats, stream, &err5);
        if (err5) return err5;
        const int OH5 = H / 2, OW5 = W / 2;
        if ((long long)N * l.c.cout_tot * mcvc_dyp_plane(OH5, OW5) > ex5.wino_cap) return MCVC_ERR_WORKSPACE;
        ex5.fail(mcvc_dyp_from_dense_launch(dy, ex5.wm, N, l.c.cout_tot, OH5, OW5, ex5.s));
        const long long dxt = (long long)N * Cin * H * W;
        int ns5 = 1;
        conv_dgrad_igemm(ex5, l.c, packed, N, H, W, ex5.wm, View{dx, (long long)Cin * H * W, (long long)H * W, W}, dxt, 0, 1, &ns5);
        if (ns5 > 1) act_fwd(ex5, dx, dxt, ns5, nullptr, 1, 1, (int)dxt, ACT_NONE);
        return ex5.err;
    }
    SchemeGuard sg(scheme);
    int err = 0;
    Exec ex = layer_exec(l.c, N, H, W, scheme, scratch, scratch_floats, stream, &err);
    if (err) return err;
    const float* params[4] = {w0, nullptr, w1, nullptr};
    ex.params = (w0 && (branches == 1 || w1)) ? params : nullptr;         // (the staged-GEMM data gradient multiplies the OIHW tensors themselves)
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    const long long dx_total = (long long)N * Cin * H * W;
    int ns = 1;
    conv_dgrad(ex, l.c, packed, N, H, W, CView{dy, (long long)l.c.cout_tot * OH * OW, (long long)OH * OW, OW},
               View{dx, (long long)Cin * H * W, (long long)H * W, W}, dx_total, 0, 1, &ns);
    if (ns > 1) act_fwd(ex, dx, dx_total, ns, nullptr, 1, 1, (int)dx_total, ACT_NONE);
    return ex.err;
}

int mcvc_layer_wgrad(const float* x, const float* dy, float* dw0, float* dw1, float* scratch, long long scratch_floats, int N, int H, int W,
                     int Cin, int Cout, int branches, int KH, int KW, int stride, int pad_h, int pad_w, int scheme, void* stream)
{
    if (!x || !dy || !dw0 || !scratch || scheme < 0 || scheme > 5 || branches < 1 || branches > 2 || (branches == 2 && !dw1)) return MCVC_ERR_INVALID;
    const LayerCtx l = layer_ctx(Cin, Cout, branches, KH, KW, stride, pad_h, pad_w);
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    if (scheme == 5) {          // implicit weight gradient (wgemm_kernels.hip): the dense input is converted to the phase-split padded layout first
        if (!igemm_applies(l.c, H, W)) return MCVC_ERR_INVALID;
        int err5 = 0;
        Exec ex5 = layer_exec(l.c, N, H, W, 0, scratch, scratch_floats, stream, &err5);
        if (err5) return err5;
        if ((long long)N * mcvc_xs_floats(Cin, H, W) > ex5.wino_cap) return MCVC_ERR_WORKSPACE;
        ex5.fail(mcvc_xs_from_dense_launch(x, ex5.wv, N, Cin, H, W, ex5.s));
        float* grads5[4] = {dw0, nullptr, dw1, nullptr};
        const long long xpl = mcvc_xs_plane(H, W);
        ex5.wgrad_x_xs = 1;
        conv_wgrad(ex5, l.c, grads5, N, H, W, CView{ex5.wv, (long long)Cin * 4 * xpl, 4 * xpl, W},
                   CView{dy, (long long)l.c.cout_tot * OH * OW, (long long)OH * OW, OW});
        return ex5.err;
    }
    SchemeGuard sg(scheme);
    int err = 0;
    Exec ex = layer_exec(l.c, N, H, W, scheme, scratch, scratch_floats, stream, &err);
    if (err) return err;
    float* grads[4] = {dw0, nullptr, dw1, nullptr};
    conv_wgrad(ex, l.c, grads, N, H, W, CView{x, (long long)Cin * H * W, (long long)H * W, W},
               CView{dy, (long long)l.c.cout_tot * OH * OW, (long long)OH * OW, OW});
    return ex.err;
}

// The fused backward of one 1-D trunk layer (resblock1d_bwd / gemm1x1_in_bwd; reference model.py:47-76 under autograd):
// InstanceNorm (+ gated GLU) backward of dy recomputed inside the transposed-convolution launch (trunk_layer_kernel<.., PRE>), d(gamma), d(beta)
// accumulated, dconv = the gradient w.r.t. the conv output stored, dx += the data gradient, and (x_in != NULL) the weight gradients by the
// batched small-K kernel.  Trunk layout [C][B][T4] for every tensor; gate pointers NULL = plain InstanceNorm (no GLU).
int mcvc_trunk_layer_backward(const float* dy, const float* conv_out, const float* stats, const float* gamma, const float* beta,
                              const float* gamma_gate, const float* beta_gate, const float* w, const float* w_gate, const float* x_in,
                              float* dx, float* dconv, float* dgamma, float* dbeta, float* dgamma_gate, float* dbeta_gate, float* dw, float* dw_gate,
                              float* wpack, int B, int Cin, int T4, int Cout, int KW, void* stream)
{
    if (!dy || !conv_out || !stats || !gamma || !beta || !w || !dx || !dconv || !wpack || (KW != 1 && KW != 3)) return MCVC_ERR_INVALID;
    const bool glu = w_gate != nullptr;
    if (glu && (!gamma_gate || !beta_gate)) return MCVC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int Cx = glu ? 2 * Cout : Cout;
    int rc = mcvc_pack_trunk_t_launch(w, wpack, Cout, Cin, KW, Cx * KW, 0, s);
    if (!rc && glu) rc = mcvc_pack_trunk_t_launch(w_gate, wpack, Cout, Cin, KW, Cx * KW, Cout, s);
    if (rc) return rc;
    const int ks = trunk_pick_ksplit(Cx, KW, Cin, B, T4, 0);
    if (ks < 1) return MCVC_ERR_INVALID;
    TrunkArgs a{};
    a.a0 = wpack;
    a.x = dy; a.x_sc = (long long)B * T4; a.x_sb = T4;
    a.Cin = Cx; a.KW = KW; a.K = Cx * KW; a.M = Cin; a.Mtot = Cin; a.B = B; a.T4 = T4; a.N = B * T4;
    a.conv_out = dx; a.c_sc = (long long)B * T4; a.c_sb = T4; a.accumulate = 1; a.mode = TRUNK_PLAIN;
    a.pre = glu ? 2 : 1; a.pre_C = Cout; a.pre_x = conv_out; a.pre_stats = stats;
    a.pre_gamma0 = gamma; a.pre_beta0 = beta; a.pre_gamma1 = gamma_gate; a.pre_beta1 = beta_gate;
    a.pre_out = dconv; a.pre_dgamma0 = dgamma; a.pre_dbeta0 = dbeta; a.pre_dgamma1 = dgamma_gate; a.pre_dbeta1 = dbeta_gate;
    rc = mcvc_trunk_launch(a, ks, s);
    if (rc || !x_in || !dw) return rc;
    if (KW != 3 || !mcvc_wgrad_smallk_batch_applies(B, T4)) return MCVC_ERR_INVALID;
    SmallKJob jobs[2];
    int nj = 0;
    jobs[nj++] = SmallKJob{x_in, dconv, dw, Cin, Cout, 0};
    if (glu && dw_gate) jobs[nj++] = SmallKJob{x_in, dconv + (long long)Cout * B * T4, dw_gate, Cin, Cout, 0};
    return mcvc_wgrad_smallk_batch_launch(jobs, nj, B, T4, s);
}

int mcvc_instnorm_act_forward(float* x, const float* gamma, const float* beta, const float* gamma_gate, const float* beta_gate,
                              const float* residual, float* y, float* stats, int N, int C, int H, int W, int act, void* stream)
{
    Exec ex{}; ex.s = (hipStream_t)stream;
    NormP np{}; np.g[0] = gamma; np.b[0] = beta; np.g[1] = gamma_gate; np.b[1] = beta_gate;
    const int Cx = (act == ACT_GLU) ? 2 * C : C;
    norm_fwd(ex, x, (long long)Cx * H * W, (long long)H * W, 0, 1, np, stats, y, (long long)C * H * W, (long long)H * W, W, residual, N, C, H, W, act);
    return ex.err;
}

int mcvc_instnorm_act_backward(const float* x, const float* gamma, const float* beta, const float* gamma_gate, const float* beta_gate,
                               const float* stats, float* dy, float* dx, float* dgamma, float* dbeta, float* dgamma_gate, float* dbeta_gate,
                               int N, int C, int H, int W, int act, void* stream)
{
    Exec ex{}; ex.s = (hipStream_t)stream;
    NormP np{}; np.g[0] = gamma; np.b[0] = beta; np.g[1] = gamma_gate; np.b[1] = beta_gate;
    np.dg[0] = dgamma; np.db[0] = dbeta; np.dg[1] = dgamma_gate; np.db[1] = dbeta_gate;
    const int Cx = (act == ACT_GLU) ? 2 * C : C;
    norm_bwd(ex, x, (long long)Cx * H * W, (long long)H * W, np, stats, dy, (long long)C * H * W, (long long)H * W, W, 0, 1,
             dx, (long long)Cx * H * W, (long long)H * W, W, 0, N, C, H, W, act);
    return ex.err;
}

// fused 1-D trunk layer (small batch): conv1d(k = 1 or 3, pad (k-1)/2) + bias + InstanceNorm1d(affine) + {gated GLU | + residual | nothing}
int mcvc_trunk_layer_forward(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, const float* w_gate,
                             const float* bias_gate, const float* gamma_gate, const float* beta_gate, const float* residual, float* conv_out,
                             float* stats, float* y, int B, int Cin, int T4, int Cout, int KW, void* stream)
{
    if (!x || !w || !bias || !gamma || !beta || !conv_out || !stats || !y) return MCVC_ERR_INVALID;
    const int mode = w_gate ? TRUNK_IN_GLU : TRUNK_IN;
    if (!mcvc_trunk_applies(Cin, KW, Cout, B, T4, mode, 1)) return MCVC_ERR_INVALID;
    TrunkArgs a{};
    a.a0 = w; a.bias0 = bias; a.gamma0 = gamma; a.beta0 = beta;
    a.a1 = w_gate; a.bias1 = bias_gate; a.gamma1 = gamma_gate; a.beta1 = beta_gate;
    a.x = x; a.x_sc = (long long)B * T4; a.x_sb = T4;
    a.Cin = Cin; a.KW = KW; a.K = Cin * KW; a.M = Cout; a.Mtot = w_gate ? 2 * Cout : Cout; a.B = B; a.T4 = T4; a.N = B * T4;
    a.conv_out = conv_out; a.c_sc = (long long)B * T4; a.c_sb = T4;
    a.stats = stats; a.y = y; a.res = residual; a.y_sn = T4; a.y_sc = (long long)B * T4; a.eps = kInEps; a.mode = mode;
    return mcvc_trunk_launch(a, 1, (hipStream_t)stream);
}

// batched fp32 GEMM of the Winograd paths: C[x] = A[x]^T-major product, A K-major ([K][lda] rows of M), B [K][ldb], C [M][ldc]
int mcvc_batched_gemm(const float* a, const float* b, float* c, int nbatch, int M, int N, int K, int lda, int ldb, int ldc,
                      long long a_stride, long long b_stride, long long c_stride, void* stream)
{
    if (!a || !b || !c || nbatch < 1) return MCVC_ERR_INVALID;
    WinoGemmArgs ga{};
    ga.a = a; ga.a_xi = a_stride; ga.lda = lda; ga.b = b; ga.b_xi = b_stride; ga.ldb = ldb; ga.c = c; ga.c_xi = c_stride; ga.ldc = ldc;
    ga.M = M; ga.N = N; ga.K = K; ga.nxi = nbatch;
    return mcvc_wino_gemm_launch(ga, (hipStream_t)stream);
}

int mcvc_bias_grad(const float* dy, float* db, int N, int C, int P, void* stream)
{
    return mcvc_bias_grad_launch(dy, (long long)C * P, (long long)P, N, C, P, db, (hipStream_t)stream);
}

int mcvc_act_forward(float* x, float* y, int N, int C, int P, int act, void* stream)
{
    Exec ex{}; ex.s = (hipStream_t)stream;
    act_fwd(ex, x, 0, 1, y, N, C, P, act);
    return ex.err;
}

int mcvc_act_backward(const float* x, float* dy, float* dx, int N, int C, int P, int act, void* stream)
{
    Exec ex{}; ex.s = (hipStream_t)stream;
    act_bwd(ex, x, dy, 0, 1, dx, N, C, P, act);
    return ex.err;
}

int mcvc_fif_input(const float* x, const float* mask, float* xin, int N, int P, void* stream)
{
    return mcvc_prep_input_launch(x, mask, xin, N, P, (hipStream_t)stream);
}

int mcvc_fif_input_grad(const float* dxin, const float* mask, float* dx, int N, int P, int accumulate, void* stream)
{
    return mcvc_mask_grad_launch(dxin, nullptr, 0, 1, mask, dx, N, P, 2, accumulate, (hipStream_t)stream);
}

}  // extern "C"
