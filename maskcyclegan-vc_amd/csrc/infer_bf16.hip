// Layer schedule of the Generator's bf16 inference forward (BASELINE configs[4]; reference call site
// mask_cyclegan_vc/test.py:85-119 -> Generator.forward, model.py:239-280) on the NHWC bf16 kernels of bf16_kernels.hip,
// and its C ABI (include/mcvc.h: mcvc_gen_bf16_*).
//
// Layout decisions (all bf16, channel-innermost):
//   * conv1 (2 -> 128|128, 5x15): the 15 kernel columns are folded into the channel axis by the input-prep kernel
//     (xin[b][h][w][kw*2+ci]), so the layer is a 5x1 convolution over 32 channels -- no 16x zero padding of Cin = 2 -- and its gated GLU
//     (model.py:242, no norm in between) is fused into the conv epilogue through an interleaved [32 value | 32 gate] row order.
//   * the reference's view(B, 5120, 1, T/4) (model.py:249-251, channel = c*20 + h) is a permutation of the conv2dto1d weight's input
//     channels (h*256 + c) plus an output stride choice of the InstanceNorm kernel in front of it (the 1 x 1 layer itself runs as a
//     1 x 5 stride-5 convolution over 1024-channel "pixels": fewer, fatter pipeline stages); view(B, 256, 20, T/4) (:270-271) is
//     the mirrored permutation of conv1dto2d's output rows.
//   * the last conv (128 -> 1, 5x15) would use 1 of 32 MFMA rows: its 15 kernel columns become 15 output channels of a 5x1 conv and a
//     small kernel adds the 15 shifted planes.
//   * PixelShuffle(2) is an index mapping inside the InstanceNorm kernels that follow it (model.py:232-236).
#include "mcvc_common.h"
#include "bf16.h"
#include "../../include/mcvc.h"

namespace {

constexpr float kEps = 1e-5f;

struct LayerB {
    int kind;                      // Bf16PackKind
    int w0, w1, b0, b1;            // parameter indices (named_parameters() order); w1 / b1 = gate branch or -1
    int Cout_src, Cin_src, KH, KW_src;
    int Cin, KW, Cout_pad;         // packed geometry
    int stride, pad_h, pad_w;
    int glu_fused;
    long long off_w, off_bias;     // byte offsets in the pack
};

struct NetB {
    LayerB conv1, ds1, ds2, c2d1d, res_vg[6], res_out[6], c1d2d, up1, up2, last;
    long long off_g6, off_b6;      // permuted affine of conv1dto2dLayer_tfan
    long long off_tvg[6], off_tout[6];   // operand-order weight copies of the residual blocks for the fused layer kernel (r6)
    long long off_c2;                    // ... of conv2dto1d for bf16_c2d1d_kernel
    long long bytes;
};

static LayerB mkl(int kind, int w0, int b0, int w1, int b1, int Cout_src, int Cin_src, int KH, int KW_src, int Cin, int KW, int Cout_pad,
                  int stride, int ph, int pw, int glu_fused = 0)
{
    LayerB l{};
    l.kind = kind; l.w0 = w0; l.b0 = b0; l.w1 = w1; l.b1 = b1; l.Cout_src = Cout_src; l.Cin_src = Cin_src; l.KH = KH; l.KW_src = KW_src;
    l.Cin = Cin; l.KW = KW; l.Cout_pad = Cout_pad; l.stride = stride; l.pad_h = ph; l.pad_w = pw; l.glu_fused = glu_fused;
    return l;
}

static NetB build_net()
{
    NetB n{};
    n.conv1 = mkl(BF16_PACK_FOLD_KW, 0, 1, 2, 3, 128, 2, 5, 15, 32, 1, 256, 1, 2, 0, 1);                  // model.py:116-126, 241-242
    n.ds1 = mkl(BF16_PACK_PLAIN, 4, 5, 8, 9, 256, 128, 5, 5, 128, 5, 512, 2, 2, 2);                       // :129-133
    n.ds2 = mkl(BF16_PACK_PLAIN, 12, 13, 16, 17, 256, 256, 5, 5, 256, 5, 512, 2, 2, 2);                   // :135-139
    n.c2d1d = mkl(BF16_PACK_HC_IN, 20, 21, -1, -1, 256, 5120, 1, 1, 1024, 5, 256, 5, 0, 0);               // :142-146 (as a 1 x 5 stride-5 conv over 1024 channels, see forward())
    // The residual blocks' 1 x 3 convolutions have 8 / 16 pipeline stages of six MFMA steps: barrier-bound.  Read as [b][2 * W4 positions]
    // [Cin / 2 channels] (the same bytes) they are 1 x 6 convolutions with stride 2 and padding 2 -- position 2 * w + g holds channel group g
    // of pixel w, so the window of output w covers positions 2 * (w - 1) .. 2 * (w - 1) + 5 = (kw, g) of the three taps: half the stages,
    // twice the steps each (knob MCVC_BF16_TRUNK_FOLD for the A/B).
    static const int fold = mcvc_knob("MCVC_BF16_TRUNK_FOLD", 1);
    const int G = fold ? 2 : 1;
    for (int i = 0; i < 6; ++i) {                                                                          // :151-180
        const int b = 24 + 12 * i;
        n.res_vg[i] = mkl(BF16_PACK_PLAIN, b + 0, b + 1, b + 4, b + 5, 512, 256, 1, 3, 256 / G, 3 * G, 1024, G, 0, G);
        n.res_out[i] = mkl(BF16_PACK_PLAIN, b + 8, b + 9, -1, -1, 256, 512, 1, 3, 512 / G, 3 * G, 256, G, 0, G);
    }
    n.c1d2d = mkl(BF16_PACK_HC_OUT, 96, 97, -1, -1, 5120, 256, 1, 1, 256, 1, 5120, 1, 0, 0);              // :183-187
    n.up1 = mkl(BF16_PACK_PLAIN, 104, 105, -1, -1, 1024, 256, 5, 5, 256, 5, 1024, 1, 2, 2);               // :192-196
    n.up2 = mkl(BF16_PACK_PLAIN, 100, 101, -1, -1, 512, 256, 5, 5, 256, 5, 512, 1, 2, 2);                 // :200-204 (convLayer.* in named_parameters)
    n.last = mkl(BF16_PACK_KW_OUT, 108, 109, -1, -1, 1, 128, 5, 15, 128, 1, 32, 1, 2, 0);                 // :207-211
    long long cur = 0;
    auto take = [&](long long bytes) { const long long o = cur; cur += (bytes + 255) & ~255LL; return o; };
    LayerB* all[] = {&n.conv1, &n.ds1, &n.ds2, &n.c2d1d, &n.c1d2d, &n.up1, &n.up2, &n.last};
    auto place = [&](LayerB& l) {
        l.off_w = take(2LL * l.Cout_pad * l.KH * l.Cin * l.KW);
        l.off_bias = take(4LL * l.Cout_pad);
    };
    for (LayerB* l : all) place(*l);
    for (int i = 0; i < 6; ++i) { place(n.res_vg[i]); place(n.res_out[i]); }
    n.off_g6 = take(4LL * 5120); n.off_b6 = take(4LL * 5120);
    for (int i = 0; i < 6; ++i) { n.off_tvg[i] = take(2 * mcvc_bf16_trunk_pack_elems(256, 512, 1)); n.off_tout[i] = take(2 * mcvc_bf16_trunk_pack_elems(512, 256, 0)); }
    n.off_c2 = take(2 * mcvc_bf16_c2d1d_pack_elems());
    n.bytes = cur;
    return n;
}
static const NetB& net() { static const NetB n = build_net(); return n; }

static inline int conv_out(int H, int K, int s, int p) { return (H + 2 * p - K) / s + 1; }

struct Dims { int B, T, W2, W4, Wu1, Wu2; };
static Dims dims(int B, int T)
{
    Dims d{};
    d.B = B; d.T = T; d.W2 = conv_out(T, 5, 2, 2); d.W4 = conv_out(d.W2, 5, 2, 2); d.Wu1 = 2 * d.W4; d.Wu2 = 4 * d.W4;
    return d;
}

struct Work {
    long long xin, y1, c2, y2, c3, y3, c4, h[2], ca, ya, cb, c6, y6, c7, y7, c8, y8, z, partial, stats, bytes;
};
static Work work(const Dims& d)
{
    Work w{};
    long long cur = 0;
    auto take = [&](long long bytes) { const long long o = cur; cur += (bytes + 255) & ~255LL; return o; };
    const long long B = d.B;
    w.xin = take(2 * B * 80 * d.T * 32);
    w.y1 = take(2 * B * 80 * d.T * 128);
    w.c2 = take(2 * B * 40 * d.W2 * 512); w.y2 = take(2 * B * 40 * d.W2 * 256);
    w.c3 = take(2 * B * 20 * d.W4 * 512); w.y3 = take(2 * B * d.W4 * 5120);
    w.c4 = take(2 * B * d.W4 * 256); w.h[0] = take(2 * B * d.W4 * 256); w.h[1] = take(2 * B * d.W4 * 256);
    w.ca = take(2 * B * d.W4 * 1024); w.ya = take(2 * B * d.W4 * 512); w.cb = take(2 * B * d.W4 * 256);
    w.c6 = take(2 * B * d.W4 * 5120); w.y6 = take(2 * B * 20 * d.W4 * 256);
    w.c7 = take(2 * B * 20 * d.W4 * 1024); w.y7 = take(2 * B * 40 * d.Wu1 * 256);
    w.c8 = take(2 * B * 40 * d.Wu1 * 512); w.y8 = take(2 * B * 80 * d.Wu2 * 128);
    w.z = take(2 * B * 80 * d.Wu2 * 32);
    w.partial = take(4LL * B * 64 * 5120 * 2);            // [N][S <= 64][Cn <= 5120][2]
    w.stats = take(4LL * B * 5120 * 2);
    w.bytes = cur;
    return w;
}

struct Run {
    hipStream_t s; int err;
    const float* const* P;
    const unsigned char* pk;
    unsigned char* ws;
    const Work* w;
    void fail(int e) { if (!err && e) err = e; }
};

// y[n][oh][ow][co] = conv(x) (+bias)
static void conv(Run& r, const LayerB& l, const bf16_t* x, long long x_sn, int x_sh, int x_sw, int N, int H, int W, bf16_t* y, long long y_sn,
                 int y_sh, int y_sw, int Cout_store)
{
    Bf16ConvArgs a{};
    a.x = x; a.x_sn = x_sn; a.x_sh = x_sh; a.x_sw = x_sw;
    a.w = reinterpret_cast<const bf16_t*>(r.pk + l.off_w);
    a.bias = reinterpret_cast<const float*>(r.pk + l.off_bias);
    a.y = y; a.y_sn = y_sn; a.y_sh = y_sh; a.y_sw = y_sw;
    a.N = N; a.H = H; a.W = W; a.Cin = l.Cin; a.Cout = Cout_store; a.Cout_pad = l.Cout_pad;
    a.KH = l.KH; a.KW = l.KW; a.stride = l.stride; a.pad_h = l.pad_h; a.pad_w = l.pad_w;
    a.OH = conv_out(H, l.KH, l.stride, l.pad_h); a.OW = conv_out(W, l.KW, l.stride, l.pad_w);
    a.glu = l.glu_fused;
    r.fail(mcvc_bf16_conv_launch(a, r.s));
}

static void norm(Run& r, const bf16_t* x, long long x_sn, int x_sh, int x_sw, int N, int H, int W, int Cx, int shuffle, int act,
                 const float* g0, const float* b0, const float* g1, const float* b1, const bf16_t* res, bf16_t* y, long long y_sn, int y_sh,
                 int y_sw, int y_csplit = 0, int y_sc2 = 0)
{
    Bf16NormArgs a{};
    a.x = x; a.x_sn = x_sn; a.x_sh = x_sh; a.x_sw = x_sw; a.N = N; a.H = H; a.W = W; a.Cx = Cx;
    a.shuffle = shuffle; a.act = act; a.has_norm = 1;
    a.gamma[0] = g0; a.beta[0] = b0; a.gamma[1] = g1; a.beta[1] = b1;
    a.partial = reinterpret_cast<float*>(r.ws + r.w->partial); a.stats = reinterpret_cast<float*>(r.ws + r.w->stats);
    const int Cn = shuffle ? Cx / 4 : Cx;
    a.S = mcvc_bf16_norm_splits(N, H * W, Cn);
    a.res = res; a.y = y; a.y_sn = y_sn; a.y_sh = y_sh; a.y_sw = y_sw; a.y_csplit = y_csplit; a.y_sc2 = y_sc2; a.eps = kEps;
    r.fail(mcvc_bf16_norm_launch(a, r.s));
}

static void forward(Run& r, const float* x, const float* mask, float* out, const Dims& d)
{
    const NetB& n = net();
    const Work& w = *r.w;
    const float* const* P = r.P;
    auto B16 = [&](long long off) { return reinterpret_cast<bf16_t*>(r.ws + off); };
    const int B = d.B, T = d.T, W2 = d.W2, W4 = d.W4, Wu1 = d.Wu1, Wu2 = d.Wu2;
    // model.py:241-242: stack(x*mask, mask), the 15 kernel columns folded into the channel axis; gated 5x15 conv, GLU in the epilogue
    // (r6: one launch -- the folded tensor is built in LDS, the weights live in registers; MCVC_BF16_CONV1_FUSED=0 in the experiments build
    //  restores input-prep kernel + generic tile for the A/B)
    static const int c1_fused = mcvc_knob("MCVC_BF16_CONV1_FUSED", 1);
    if (c1_fused) {
        r.fail(mcvc_bf16_conv1_fused_launch(x, mask, reinterpret_cast<const bf16_t*>(r.pk + n.conv1.off_w), reinterpret_cast<const float*>(r.pk + n.conv1.off_bias),
                                            B16(w.y1), B, 80, T, r.s));
    } else {
        r.fail(mcvc_bf16_prep_launch(x, mask, B16(w.xin), B, 80, T, r.s));
        conv(r, n.conv1, B16(w.xin), 80LL * T * 32, T * 32, 32, B, 80, T, B16(w.y1), 80LL * T * 128, T * 128, 128, 128);
    }
    // :245 downSample1
    conv(r, n.ds1, B16(w.y1), 80LL * T * 128, T * 128, 128, B, 80, T, B16(w.c2), 40LL * W2 * 512, W2 * 512, 512, 512);
    norm(r, B16(w.c2), 40LL * W2 * 512, W2 * 512, 512, B, 40, W2, 512, 0, BF16_ACT_GLU, P[6], P[7], P[10], P[11], nullptr,
         B16(w.y2), 40LL * W2 * 256, W2 * 256, 256);
    // :246 downSample2; output written as [b][w][h*256 + c]  (:249-251)
    conv(r, n.ds2, B16(w.y2), 40LL * W2 * 256, W2 * 256, 256, B, 40, W2, B16(w.c3), 20LL * W4 * 512, W4 * 512, 512, 512);
    norm(r, B16(w.c3), 20LL * W4 * 512, W4 * 512, 512, B, 20, W4, 512, 0, BF16_ACT_GLU, P[14], P[15], P[18], P[19], nullptr,
         B16(w.y3), (long long)W4 * 5120, 256, 5120);
    // :254-255 1x1 5120 -> 256 + IN.  K = 5120 in 32-channel stages of two MFMA steps each is 160 barrier-bound stages per workgroup; the
    // same bytes read as [b][5 * W4 "pixels"][1024 channels] make it a 1 x 5 convolution with stride 5 (non-overlapping windows): 32 stages
    // of ten steps, the weight pack groups the channels accordingly (r5: 122 -> see DESIGN 8b)
    // (r6) T <= 512 frames: conv + norm in ONE launch (operand-order weights, X through a four-stage LDS-DMA ring); MCVC_BF16_C2D1D_FUSED=0 in the
    // experiments build restores the two-launch form
    static const int c2_fused = mcvc_knob("MCVC_BF16_C2D1D_FUSED", 1);
    if (c2_fused && mcvc_bf16_c2d1d_applies(W4)) {
        r.fail(mcvc_bf16_c2d1d_launch(B16(w.y3), (long long)W4 * 5120, reinterpret_cast<const bf16_t*>(r.pk + n.off_c2), P[22], P[23], B16(w.h[0]),
                                      (long long)W4 * 256, B, W4, kEps, r.s));
    } else {
        conv(r, n.c2d1d, B16(w.y3), (long long)W4 * 5120, 0, 1024, B, 1, 5 * W4, B16(w.c4), (long long)W4 * 256, 0, 256, 256);
        norm(r, B16(w.c4), (long long)W4 * 256, 0, 256, B, 1, W4, 256, 0, BF16_ACT_NONE, P[22], P[23], nullptr, nullptr, nullptr,
             B16(w.h[0]), (long long)W4 * 256, 0, 256);
    }
    // :258-263 residual blocks
    int cur = 0;
    // (r6) T <= 512 frames: a sample's row is one tile wide, conv + InstanceNorm + GLU / residual of a layer are ONE launch (12 launches for the
    // six blocks instead of 24); longer inputs, or MCVC_BF16_TRUNK_FUSED=0 in the experiments build, take the two-launch form below
    static const int trunk_fused = mcvc_knob("MCVC_BF16_TRUNK_FUSED", 1);
    const bool fused = trunk_fused && mcvc_bf16_trunk_layer_applies(W4, 256, 512) && mcvc_bf16_trunk_layer_applies(W4, 512, 256);
    for (int i = 0; i < 6 && fused; ++i) {
        const int b = 24 + 12 * i;
        r.fail(mcvc_bf16_trunk_layer_launch(B16(w.h[cur]), (long long)W4 * 256, reinterpret_cast<const bf16_t*>(r.pk + n.off_tvg[i]), P[b + 2], P[b + 3], P[b + 6],
                                            P[b + 7], nullptr, B16(w.ya), (long long)W4 * 512, B, W4, 256, 512, 1, kEps, r.s));
        r.fail(mcvc_bf16_trunk_layer_launch(B16(w.ya), (long long)W4 * 512, reinterpret_cast<const bf16_t*>(r.pk + n.off_tout[i]), P[b + 10], P[b + 11], nullptr,
                                            nullptr, B16(w.h[cur]), B16(w.h[cur ^ 1]), (long long)W4 * 256, B, W4, 512, 256, 0, kEps, r.s));
        cur ^= 1;
    }
    for (int i = 0; i < 6 && !fused; ++i) {
        const int b = 24 + 12 * i;
        const int G = n.res_vg[i].stride;                   // channel groups folded into the position axis (build_net)
        conv(r, n.res_vg[i], B16(w.h[cur]), (long long)W4 * 256, 0, 256 / G, B, 1, G * W4, B16(w.ca), (long long)W4 * 1024, 0, 1024, 1024);
        norm(r, B16(w.ca), (long long)W4 * 1024, 0, 1024, B, 1, W4, 1024, 0, BF16_ACT_GLU, P[b + 2], P[b + 3], P[b + 6], P[b + 7], nullptr,
             B16(w.ya), (long long)W4 * 512, 0, 512);
        conv(r, n.res_out[i], B16(w.ya), (long long)W4 * 512, 0, 512 / G, B, 1, G * W4, B16(w.cb), (long long)W4 * 256, 0, 256, 256);
        norm(r, B16(w.cb), (long long)W4 * 256, 0, 256, B, 1, W4, 256, 0, BF16_ACT_NONE, P[b + 10], P[b + 11], nullptr, nullptr, B16(w.h[cur]),
             B16(w.h[cur ^ 1]), (long long)W4 * 256, 0, 256);
        cur ^= 1;
    }
    // :266-271 1x1 256 -> 5120 (rows h*256 + c) + IN, written NHWC [b][h][w][c]
    conv(r, n.c1d2d, B16(w.h[cur]), (long long)W4 * 256, 0, 256, B, 1, W4, B16(w.c6), (long long)W4 * 5120, 0, 5120, 5120);
    norm(r, B16(w.c6), (long long)W4 * 5120, 0, 5120, B, 1, W4, 5120, 0, BF16_ACT_NONE, reinterpret_cast<const float*>(r.pk + n.off_g6),
         reinterpret_cast<const float*>(r.pk + n.off_b6), nullptr, nullptr, nullptr, B16(w.y6), 20LL * W4 * 256, 0, 256, 256, W4 * 256);
    // :274 upSample1: conv -> PixelShuffle -> IN -> x*sigmoid(x)
    conv(r, n.up1, B16(w.y6), 20LL * W4 * 256, W4 * 256, 256, B, 20, W4, B16(w.c7), 20LL * W4 * 1024, W4 * 1024, 1024, 1024);
    norm(r, B16(w.c7), 20LL * W4 * 1024, W4 * 1024, 1024, B, 20, W4, 1024, 1, BF16_ACT_SILU, P[106], P[107], nullptr, nullptr, nullptr,
         B16(w.y7), 40LL * Wu1 * 256, Wu1 * 256, 256);
    // :275 upSample2
    conv(r, n.up2, B16(w.y7), 40LL * Wu1 * 256, Wu1 * 256, 256, B, 40, Wu1, B16(w.c8), 40LL * Wu1 * 512, Wu1 * 512, 512, 512);
    norm(r, B16(w.c8), 40LL * Wu1 * 512, Wu1 * 512, 512, B, 40, Wu1, 512, 1, BF16_ACT_SILU, P[102], P[103], nullptr, nullptr, nullptr,
         B16(w.y8), 80LL * Wu2 * 128, Wu2 * 128, 128);
    // :278-279 last conv: 15 kernel columns as output channels, then the column sum (+ bias)
    // (r6: one launch, fp32 kernel-column sum in LDS; MCVC_BF16_LAST_FUSED=0 in the experiments build restores conv + shifted-plane sum)
    static const int last_fused = mcvc_knob("MCVC_BF16_LAST_FUSED", 1);
    if (last_fused) {
        r.fail(mcvc_bf16_last_fused_launch(B16(w.y8), reinterpret_cast<const bf16_t*>(r.pk + n.last.off_w), P[109], out, B, 80, Wu2, r.s));
    } else {
        conv(r, n.last, B16(w.y8), 80LL * Wu2 * 128, Wu2 * 128, 128, B, 80, Wu2, B16(w.z), 80LL * Wu2 * 32, Wu2 * 32, 32, 32);
        r.fail(mcvc_bf16_last_launch(B16(w.z), P[109], out, B, 80, Wu2, r.s));
    }
}

static void pack_layer(Run& r, const LayerB& l, unsigned char* pk)
{
    Bf16PackArgs a{};
    a.w[0] = r.P[l.w0]; a.w[1] = l.w1 >= 0 ? r.P[l.w1] : nullptr;
    a.dst = reinterpret_cast<bf16_t*>(pk + l.off_w);
    a.kind = l.kind; a.nbr = l.w1 >= 0 ? 2 : 1; a.Cout_src = l.Cout_src; a.Cin_src = l.Cin_src; a.KH = l.KH; a.KW_src = l.KW_src;
    a.Cout_pad = l.Cout_pad; a.KW = l.KW; a.Cin = l.Cin; a.glu_interleave = l.glu_fused;
    r.fail(mcvc_bf16_pack_launch(a, r.s));
    float* bias = reinterpret_cast<float*>(pk + l.off_bias);
    if (l.kind == BF16_PACK_KW_OUT) r.fail(mcvc_bf16_vec_launch(r.P[l.b0], nullptr, bias, 0, l.Cout_pad, BF16_PACK_PLAIN, r.s));   // zeros: bias added after the column sum
    else if (l.glu_fused) r.fail(mcvc_bf16_vec_launch(r.P[l.b0], r.P[l.b1], bias, l.Cout_src, l.Cout_pad, -1, r.s));
    else if (l.kind == BF16_PACK_HC_OUT) r.fail(mcvc_bf16_vec_launch(r.P[l.b0], nullptr, bias, l.Cout_src, l.Cout_pad, BF16_PACK_HC_OUT, r.s));
    else r.fail(mcvc_bf16_vec_launch(r.P[l.b0], l.b1 >= 0 ? r.P[l.b1] : nullptr, bias, l.Cout_src, l.Cout_pad, BF16_PACK_PLAIN, r.s));
}

}  // namespace

extern "C" {

long long mcvc_gen_bf16_packed_bytes(void) { return net().bytes; }

long long mcvc_gen_bf16_workspace_bytes(int B, int T)
{
    if (B < 1 || T < 1) return -1;
    return work(dims(B, T)).bytes;
}

int mcvc_gen_bf16_pack(const float* const* params, void* packed, void* stream)
{
    if (!params || !packed) return MCVC_ERR_INVALID;
    Run r{}; r.s = (hipStream_t)stream; r.P = params;
    unsigned char* pk = static_cast<unsigned char*>(packed);
    const NetB& n = net();
    const LayerB* all[] = {&n.conv1, &n.ds1, &n.ds2, &n.c2d1d, &n.c1d2d, &n.up1, &n.up2, &n.last};
    for (const LayerB* l : all) pack_layer(r, *l, pk);
    for (int i = 0; i < 6; ++i) { pack_layer(r, n.res_vg[i], pk); pack_layer(r, n.res_out[i], pk); }
    r.fail(mcvc_bf16_c2d1d_pack_launch(params[20], reinterpret_cast<bf16_t*>(pk + n.off_c2), r.s));
    for (int i = 0; i < 6; ++i) {
        const int b = 24 + 12 * i;
        r.fail(mcvc_bf16_trunk_pack_launch(params[b + 0], params[b + 4], reinterpret_cast<bf16_t*>(pk + n.off_tvg[i]), 256, 512, 1, r.s));
        r.fail(mcvc_bf16_trunk_pack_launch(params[b + 8], nullptr, reinterpret_cast<bf16_t*>(pk + n.off_tout[i]), 512, 256, 0, r.s));
    }
    r.fail(mcvc_bf16_vec_launch(params[98], nullptr, reinterpret_cast<float*>(pk + n.off_g6), 5120, 5120, BF16_PACK_HC_OUT, r.s));
    r.fail(mcvc_bf16_vec_launch(params[99], nullptr, reinterpret_cast<float*>(pk + n.off_b6), 5120, 5120, BF16_PACK_HC_OUT, r.s));
    return r.err;
}

int mcvc_gen_infer_bf16(const float* const* params, const void* packed, const float* x, const float* mask, float* out, void* workspace,
                        long long workspace_bytes, int B, int T, void* stream)
{
    if (B < 1 || T < 1 || !params || !packed || !x || !out || !workspace) return MCVC_ERR_INVALID;
    const Dims d = dims(B, T);
    if (d.W4 < 1) return MCVC_ERR_INVALID;
    const Work w = work(d);
    if (workspace_bytes < w.bytes) return MCVC_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(packed)) & 15) return MCVC_ERR_INVALID;
    Run r{}; r.s = (hipStream_t)stream; r.P = params; r.pk = static_cast<const unsigned char*>(packed);
    r.ws = static_cast<unsigned char*>(workspace); r.w = &w;
    forward(r, x, mask, out, d);
    return r.err;
}

// ---- single-op entry points (kernel parity tests; same kernels the forward uses) -----------------------------------------------
// y[N][OH][OW][Cout] (bf16 NHWC) = conv2d(x[N][H][W][Cin] bf16 NHWC, w[Cout][Cin][KH][KW] fp32 OIHW -> bf16) + bias
long long mcvc_bf16_conv2d_pack_bytes(int Cout, int Cin, int KH, int KW)
{
    const int cout_pad = (Cout % 128 == 0) ? Cout : ((Cout + 31) / 32) * 32;
    return 2LL * cout_pad * KH * Cin * KW + 4LL * cout_pad + 512;
}

int mcvc_bf16_conv2d(const void* x, const float* w, const float* bias, void* y, void* wpack, int N, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad_h, int pad_w, void* stream)
{
    if ((Cin & 31) || (Cout & 3) || !x || !w || !y || !wpack) return MCVC_ERR_INVALID;
    const int cout_pad = (Cout % 128 == 0) ? Cout : ((Cout + 31) / 32) * 32;
    LayerB l = mkl(BF16_PACK_PLAIN, 0, 1, -1, -1, Cout, Cin, KH, KW, Cin, KW, cout_pad, stride, pad_h, pad_w);
    l.off_w = 0; l.off_bias = ((2LL * cout_pad * KH * Cin * KW) + 255) & ~255LL;
    const float* P[2] = {w, bias};
    Run r{}; r.s = (hipStream_t)stream; r.P = P; r.pk = static_cast<const unsigned char*>(wpack);
    Bf16PackArgs a{};
    a.w[0] = w; a.dst = reinterpret_cast<bf16_t*>(wpack); a.kind = BF16_PACK_PLAIN; a.nbr = 1; a.Cout_src = Cout; a.Cin_src = Cin; a.KH = KH; a.KW_src = KW;
    a.Cout_pad = cout_pad; a.KW = KW; a.Cin = Cin;
    r.fail(mcvc_bf16_pack_launch(a, r.s));
    r.fail(mcvc_bf16_vec_launch(bias, nullptr, reinterpret_cast<float*>(static_cast<unsigned char*>(wpack) + l.off_bias), bias ? Cout : 0, cout_pad,
                                BF16_PACK_PLAIN, r.s));
    const int OH = conv_out(H, KH, stride, pad_h), OW = conv_out(W, KW, stride, pad_w);
    conv(r, l, static_cast<const bf16_t*>(x), (long long)H * W * Cin, W * Cin, Cin, N, H, W, static_cast<bf16_t*>(y), (long long)OH * OW * Cout,
         OW * Cout, Cout, Cout);
    return r.err;
}

// y[B][80][T][128] (bf16 NHWC) = conv1(stack(x * mask, mask)) * sigmoid(conv1_gates(...)) -- model.py:241-242 -- through the fused kernel the
// forward uses: w / wg [128][2][5][15], b / bg [128] fp32; x, mask fp32 [B][80][T] (mask NULL = ones); wpack: mcvc_bf16_conv1_glu_pack_bytes()
long long mcvc_bf16_conv1_glu_pack_bytes(void) { return 2LL * 256 * 5 * 32 + 4LL * 256 + 512; }

int mcvc_bf16_conv1_glu(const float* x, const float* mask, const float* w, const float* b, const float* wg, const float* bg, void* y, void* wpack,
                        int B, int T, void* stream)
{
    if (!x || !w || !b || !wg || !bg || !y || !wpack || B < 1 || T < 1) return MCVC_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(wpack) & 15) return MCVC_ERR_INVALID;
    LayerB l = net().conv1;
    l.w0 = 0; l.b0 = 1; l.w1 = 2; l.b1 = 3;
    l.off_w = 0; l.off_bias = (2LL * 256 * 5 * 32 + 255) & ~255LL;
    const float* P[4] = {w, b, wg, bg};
    Run r{}; r.s = (hipStream_t)stream; r.P = P;
    unsigned char* pk = static_cast<unsigned char*>(wpack);
    pack_layer(r, l, pk);
    r.fail(mcvc_bf16_conv1_fused_launch(x, mask, reinterpret_cast<const bf16_t*>(pk + l.off_w), reinterpret_cast<const float*>(pk + l.off_bias),
                                        static_cast<bf16_t*>(y), B, 80, T, r.s));
    return r.err;
}

// out[B][80][T] fp32 = conv2d(x, w, b, padding (2, 7)) for the generator's last layer (model.py:207-211): x [B][80][T][128] bf16 NHWC,
// w [1][128][5][15], b [1] fp32 -- through the fused kernel the forward uses; wpack: mcvc_bf16_last_conv_pack_bytes()
long long mcvc_bf16_last_conv_pack_bytes(void) { return 2LL * 32 * 5 * 128 + 4LL * 32 + 512; }

int mcvc_bf16_last_conv(const void* x, const float* w, const float* b, float* out, void* wpack, int B, int T, void* stream)
{
    if (!x || !w || !out || !wpack || B < 1 || T < 1) return MCVC_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(wpack) & 15) return MCVC_ERR_INVALID;
    LayerB l = net().last;
    l.w0 = 0; l.b0 = 1;
    l.off_w = 0; l.off_bias = (2LL * 32 * 5 * 128 + 255) & ~255LL;
    const float* P[2] = {w, b};
    Run r{}; r.s = (hipStream_t)stream; r.P = P;
    unsigned char* pk = static_cast<unsigned char*>(wpack);
    pack_layer(r, l, pk);
    r.fail(mcvc_bf16_last_fused_launch(static_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(pk + l.off_w), b, out, B, 80, T, r.s));
    return r.err;
}

// conv2dto1d + its InstanceNorm through the fused kernel (model.py:142-146, 254-255): x [B][W][5120] bf16 with memory channel h * 256 + c = the
// reference's channel c * 20 + h (how the forward lays out downSample2's output), w [256][5120] fp32 in the REFERENCE's channel order, gamma /
// beta [256]; y [B][W][256] bf16; W <= 128; wpack: mcvc_bf16_c2d1d_pack_bytes()
long long mcvc_bf16_c2d1d_pack_bytes(void) { return 2 * mcvc_bf16_c2d1d_pack_elems() + 256; }

int mcvc_bf16_c2d1d_norm(const void* x, const float* w, const float* gamma, const float* beta, void* y, void* wpack, int B, int W, void* stream)
{
    if (!x || !w || !gamma || !beta || !y || !wpack || B < 1 || !mcvc_bf16_c2d1d_applies(W)) return MCVC_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(wpack) & 15) return MCVC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int e = mcvc_bf16_c2d1d_pack_launch(w, static_cast<bf16_t*>(wpack), s);
    if (e) return e;
    return mcvc_bf16_c2d1d_launch(static_cast<const bf16_t*>(x), (long long)W * 5120, static_cast<const bf16_t*>(wpack), gamma, beta, static_cast<bf16_t*>(y),
                                  (long long)W * 256, B, W, kEps, s);
}

// one residual-block layer through the fused kernel (model.py:47-76): x [B][W][Cin] bf16, w / w_gate [C][Cin][3] fp32 (w_gate NULL: plain layer with
// optional residual [B][W][C]), gamma / beta (+ gate's) [C]; y [B][W][C] bf16; wpack: mcvc_bf16_trunk_layer_pack_bytes(Cin, C, gated) bytes
long long mcvc_bf16_trunk_layer_pack_bytes(int Cin, int C, int gated) { return 2 * mcvc_bf16_trunk_pack_elems(Cin, C, gated) + 256; }

int mcvc_bf16_trunk_layer(const void* x, const float* w, const float* w_gate, const float* gamma, const float* beta, const float* gamma_gate,
                          const float* beta_gate, const void* residual, void* y, void* wpack, int B, int W, int Cin, int C, void* stream)
{
    if (!x || !w || !gamma || !beta || !y || !wpack || B < 1) return MCVC_ERR_INVALID;
    if (w_gate && (!gamma_gate || !beta_gate)) return MCVC_ERR_INVALID;
    if (!mcvc_bf16_trunk_layer_applies(W, Cin, C) || (reinterpret_cast<uintptr_t>(wpack) & 15)) return MCVC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int e = mcvc_bf16_trunk_pack_launch(w, w_gate, static_cast<bf16_t*>(wpack), Cin, C, w_gate ? 1 : 0, s);
    if (e) return e;
    return mcvc_bf16_trunk_layer_launch(static_cast<const bf16_t*>(x), (long long)W * Cin, static_cast<const bf16_t*>(wpack), gamma, beta, gamma_gate, beta_gate,
                                        static_cast<const bf16_t*>(residual), static_cast<bf16_t*>(y), (long long)W * C, B, W, Cin, C, w_gate ? 1 : 0, kEps, s);
}

// y = act(InstanceNorm(x)) on NHWC bf16.  x: [N][H][W][Cx]; act 0 none, 1 gated GLU (Cx = 2C: value | gate), 2 x*sigmoid(x);
// pixel_shuffle != 0: the normalised tensor is PixelShuffle(2)(x) (C = Cx / 4, output [N][2H][2W][C]).  scratch: N*(64+1)*Cx*2 floats.
int mcvc_bf16_instnorm_act(const void* x, const float* gamma, const float* beta, const float* gamma_gate, const float* beta_gate,
                           const void* residual, void* y, float* scratch, int N, int H, int W, int Cx, int act, int pixel_shuffle, void* stream)
{
    if (!x || !y || !scratch || !gamma || !beta) return MCVC_ERR_INVALID;
    const int C = pixel_shuffle ? Cx / 4 : (act == BF16_ACT_GLU ? Cx / 2 : Cx);
    Bf16NormArgs a{};
    a.x = static_cast<const bf16_t*>(x); a.x_sn = (long long)H * W * Cx; a.x_sh = W * Cx; a.x_sw = Cx; a.N = N; a.H = H; a.W = W; a.Cx = Cx;
    a.shuffle = pixel_shuffle ? 1 : 0; a.act = act; a.has_norm = 1;
    a.gamma[0] = gamma; a.beta[0] = beta; a.gamma[1] = gamma_gate; a.beta[1] = beta_gate;
    const int Cn = pixel_shuffle ? Cx / 4 : Cx;
    a.S = mcvc_bf16_norm_splits(N, H * W, Cn);
    a.partial = scratch; a.stats = scratch + (long long)N * 64 * Cx * 2;
    a.res = static_cast<const bf16_t*>(residual); a.y = static_cast<bf16_t*>(y);
    const int OHn = pixel_shuffle ? 2 * H : H, OWn = pixel_shuffle ? 2 * W : W;
    a.y_sn = (long long)OHn * OWn * C; a.y_sh = OWn * C; a.y_sw = C; a.eps = kEps;
    return mcvc_bf16_norm_launch(a, (hipStream_t)stream);
}

}  // extern "C"
