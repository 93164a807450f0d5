"""GPU: the bf16 inference path (csrc/bf16_kernels.hip, infer_bf16.hip) -- BASELINE configs[4].

Single ops against a plain PyTorch fp32 reference evaluated on the SAME bf16-rounded operands (so the comparison measures the
kernel -- accumulation order and the one bf16 rounding of the output -- not the quantisation of the inputs): 4e-3 relative L2
(bf16 has 8 mantissa bits: output rounding alone is ~2.3e-3 rms).  Whole generator against the CPU oracle in fp32: gate 2e-2
relative L2, at small shapes, ragged T, and the full bs=16 x 512 frames configuration."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import mcvc_oracle as orc  # noqa: E402
from mask_cyclegan_vc import _hip  # noqa: E402
from mask_cyclegan_vc._hip import check, lib, ptr, stream  # noqa: E402
from mask_cyclegan_vc.model import Generator  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


CONV_CASES = [
    # N, H, W, Cin, Cout, KH, KW, stride, ph, pw                  what it is in the generator
    (2, 80, 64, 32, 256, 5, 1, 1, 2, 0),     # conv1 after the kw fold
    (2, 80, 64, 128, 512, 5, 5, 2, 2, 2),    # downSample1 (value | gate)
    (1, 40, 32, 256, 512, 5, 5, 2, 2, 2),    # downSample2
    (2, 1, 16, 5120, 256, 1, 1, 1, 0, 0),    # conv2dto1d
    (3, 1, 16, 256, 1024, 1, 3, 1, 0, 1),    # residual value | gate
    (3, 1, 16, 512, 256, 1, 3, 1, 0, 1),     # residual out
    (2, 1, 16, 256, 5120, 1, 1, 1, 0, 0),    # conv1dto2d
    (1, 20, 16, 256, 1024, 5, 5, 1, 2, 2),   # upSample1
    (1, 40, 32, 256, 512, 5, 5, 1, 2, 2),    # upSample2
    (1, 80, 64, 128, 32, 5, 1, 1, 2, 0),     # last conv after the kw -> channel fold (32-row tile config)
    (1, 13, 37, 64, 36, 3, 3, 1, 1, 1),      # ragged: overhanging tiles, Cout not a multiple of 32
    (1, 21, 45, 32, 128, 5, 5, 2, 2, 2),     # ragged stride 2
    (2, 1, 80, 1024, 256, 1, 5, 5, 0, 0),    # conv2dto1d as the forward runs it since r5: 1 x 5, stride 5 over [5 * W4 positions][1024 channels]
    (3, 1, 32, 128, 1024, 1, 6, 2, 0, 2),    # residual value | gate as the forward runs it: 1 x 6, stride 2, two channel groups as positions
    (3, 1, 32, 256, 256, 1, 6, 2, 0, 2),     # residual out, the same fold
    (1, 9, 31, 32, 64, 3, 3, 3, 1, 1),       # a stride the generator does not use (general stride in the pixel addressing), ragged
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,KH,KW,stride,ph,pw", CONV_CASES)
def test_bf16_conv_matches_fp32_conv_on_the_same_rounded_operands(N, H, W, Cin, Cout, KH, KW, stride, ph, pw):
    L = lib()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    OH, OW = (H + 2 * ph - KH) // stride + 1, (W + 2 * pw - KW) // stride + 1
    y = torch.empty(N, OH, OW, Cout, dtype=torch.bfloat16, device="cuda")
    wpack = torch.zeros(L.mcvc_bf16_conv2d_pack_bytes(Cout, Cin, KH, KW), dtype=torch.uint8, device="cuda")
    check(L.mcvc_bf16_conv2d(ptr(x_nhwc), ptr(w), ptr(b), ptr(y), ptr(wpack), N, H, W, Cin, Cout, KH, KW, stride, ph, pw, stream()), "bf16_conv2d")
    ref = F.conv2d(x.float(), w.to(torch.bfloat16).float(), b, stride, (ph, pw))
    got = y.float().permute(0, 3, 1, 2)
    e = rel(got, ref)
    assert e < 4e-3, e


@pytest.mark.parametrize("B,T,masked", [(2, 64, True), (1, 65, False), (3, 40, True), (1, 512, True), (2, 20, False)])
def test_bf16_fused_conv1_matches_fp32_on_the_same_rounded_operands(B, T, masked):
    """r6: conv1 + its input preparation + its gated GLU in ONE launch (bf16_conv1_fused_kernel: the (x * mask, mask) strip built in LDS from the
    fp32 inputs, the layer's weights in registers) -- model.py:241-242.  Reference: F.conv2d on the bf16-rounded stack(x * mask, mask) with
    the bf16-rounded weights, fp32 bias, value * sigmoid(gate); ragged T (strips of 32 columns overhang), T < 32, mask = None (ones)."""
    L = lib()
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, 80, T, generator=g).cuda()
    mask = None
    if masked:
        mask = torch.ones(B, 80, T)
        for b in range(B):
            s0 = int(torch.randint(0, max(T - 12, 1), (1,), generator=g)); mask[b, :, s0:s0 + min(12, T // 2)] = 0.0
        mask = mask.cuda()
    w = (torch.randn(128, 2, 5, 15, generator=g) / 150 ** 0.5).cuda(); wg = (torch.randn(128, 2, 5, 15, generator=g) / 150 ** 0.5).cuda()
    bv, bg = torch.randn(128, generator=g).cuda(), torch.randn(128, generator=g).cuda()
    y = torch.full((B, 80, T, 128), float("nan"), dtype=torch.bfloat16, device="cuda")
    wpack = torch.zeros(L.mcvc_bf16_conv1_glu_pack_bytes(), dtype=torch.uint8, device="cuda")
    check(L.mcvc_bf16_conv1_glu(ptr(x), ptr(mask), ptr(w), ptr(bv), ptr(wg), ptr(bg), ptr(y), ptr(wpack), B, T, stream()), "bf16_conv1_glu")
    m = mask if mask is not None else torch.ones_like(x)
    xin = torch.stack((x * m, m), dim=1).to(torch.bfloat16).float()
    v = F.conv2d(xin, w.to(torch.bfloat16).float(), bv, 1, (2, 7))
    gt = F.conv2d(xin, wg.to(torch.bfloat16).float(), bg, 1, (2, 7))
    ref = v * torch.sigmoid(gt)
    got = y.float().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    e = rel(got, ref)
    assert e < 4e-3, e


@pytest.mark.parametrize("B,W", [(3, 16), (16, 128), (1, 17), (9, 100), (2, 2)])
def test_bf16_fused_conv2dto1d_norm_matches_fp32_on_the_same_rounded_operands(B, W):
    """r6: conv2dto1d (5120 -> 256, k = 1) + its InstanceNorm in ONE launch (bf16_c2d1d_kernel: X through a four-stage LDS-DMA ring with swizzled
    16-byte pieces, operand-order weights in a register ring, hand-counted vmcnt waits) -- model.py:142-146, 249-255.  The kernel reads the
    layout the forward produces (memory channel h * 256 + c = reference channel c * 20 + h) and the weight in the reference's order.
    B = 9 / 16: more than one group of eight samples (the workgroup -> (sample, channel tile) map); ragged W; W = 2."""
    L = lib()
    g = torch.Generator().manual_seed(41)
    xr = torch.randn(B, 5120, W, generator=g).cuda().to(torch.bfloat16)               # the reference's channel order c * 20 + h
    w = (torch.randn(256, 5120, generator=g) / 5120 ** 0.5).cuda()
    ga, be = (1 + 0.2 * torch.randn(256, generator=g)).cuda(), (0.3 * torch.randn(256, generator=g)).cuda()
    xm = xr.view(B, 256, 20, W).permute(0, 3, 2, 1).contiguous().view(B, W, 5120)       # [b][w][h * 256 + c]
    y = torch.full((B, W, 256), float("nan"), dtype=torch.bfloat16, device="cuda")
    wpack = torch.zeros(L.mcvc_bf16_c2d1d_pack_bytes(), dtype=torch.uint8, device="cuda")
    check(L.mcvc_bf16_c2d1d_norm(ptr(xm), ptr(w), ptr(ga), ptr(be), ptr(y), ptr(wpack), B, W, stream()), "bf16_c2d1d_norm")
    z = F.instance_norm(F.conv1d(xr.float(), w.to(torch.bfloat16).float().unsqueeze(2)), weight=ga, bias=be, eps=1e-5)
    got = y.float().permute(0, 2, 1)
    assert torch.isfinite(got).all()
    assert rel(got, z) < 4e-3, rel(got, z)


@pytest.mark.parametrize("B,W,Cin,C,gated,res", [(3, 16, 256, 512, True, False), (3, 16, 512, 256, False, True), (2, 128, 256, 512, True, False),
                                                  (2, 128, 512, 256, False, True), (1, 17, 256, 512, True, False), (2, 100, 512, 256, False, False),
                                                  (1, 2, 256, 32, False, True), (2, 33, 512, 64, True, False)])
def test_bf16_fused_trunk_layer_matches_fp32_on_the_same_rounded_operands(B, W, Cin, C, gated, res):
    """r6: a residual-block layer (model.py:47-76) in ONE launch at inference -- conv1d(k = 3) + InstanceNorm1d + GLU / residual
    (bf16_trunk_layer_kernel: the sample's row block staged once in LDS, weights streamed in MFMA operand order, statistics on the fp32
    accumulators).  Reference: F.conv1d on the bf16-rounded x with bf16-rounded weights, F.instance_norm, fp32; W = 16 (64 frames), 128 (512
    frames: the benchmark shape), ragged widths, a single column."""
    L = lib()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, W, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(C, Cin, 3, generator=g) / (3 * Cin) ** 0.5).cuda()
    wg = (torch.randn(C, Cin, 3, generator=g) / (3 * Cin) ** 0.5).cuda() if gated else None
    ga, be = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()
    gg, bg = ((1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()) if gated else (None, None)
    r = torch.randn(B, W, C, generator=g).cuda().to(torch.bfloat16) if res else None
    y = torch.full((B, W, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    wpack = torch.zeros(L.mcvc_bf16_trunk_layer_pack_bytes(Cin, C, 1 if gated else 0), dtype=torch.uint8, device="cuda")
    check(L.mcvc_bf16_trunk_layer(ptr(x.permute(0, 2, 1).contiguous()), ptr(w), ptr(wg), ptr(ga), ptr(be), ptr(gg), ptr(bg), ptr(r), ptr(y), ptr(wpack),
                                  B, W, Cin, C, stream()), "bf16_trunk_layer")
    xf = x.float()
    z = F.instance_norm(F.conv1d(xf, w.to(torch.bfloat16).float(), None, 1, 1), weight=ga, bias=be, eps=1e-5)
    if gated:
        z = z * torch.sigmoid(F.instance_norm(F.conv1d(xf, wg.to(torch.bfloat16).float(), None, 1, 1), weight=gg, bias=bg, eps=1e-5))
    if res:
        z = z + r.float().permute(0, 2, 1)
    got = y.float().permute(0, 2, 1)
    assert torch.isfinite(got).all()
    assert rel(got, z) < 4e-3, rel(got, z)


@pytest.mark.parametrize("B,T", [(2, 64), (1, 65), (1, 512), (3, 20), (1, 240)])
def test_bf16_fused_last_conv_matches_fp32_on_the_same_rounded_operands(B, T):
    """r6: the generator's last conv (128 -> 1, 5 x 15; model.py:207-211) in ONE launch (bf16_last_fused_kernel: 15 kernel columns as MFMA rows,
    weights in registers, every input row loaded once for its five output rows, fp32 kernel-column sum through LDS).  Reference: F.conv2d on
    the bf16 activations with bf16-rounded weights in fp32.  The output is fp32 (no output rounding): 1e-3.  T = 65 / 240 / 20: ragged strips
    of 114 output columns, a strip narrower than one tile."""
    L = lib()
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, 128, 80, T, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(1, 128, 5, 15, generator=g) / (128 * 75) ** 0.5).cuda()
    b = torch.randn(1, generator=g).cuda()
    out = torch.full((B, 80, T), float("nan"), device="cuda")
    wpack = torch.zeros(L.mcvc_bf16_last_conv_pack_bytes(), dtype=torch.uint8, device="cuda")
    check(L.mcvc_bf16_last_conv(ptr(x.permute(0, 2, 3, 1).contiguous()), ptr(w), ptr(b), ptr(out), ptr(wpack), B, T, stream()), "bf16_last_conv")
    ref = F.conv2d(x.float(), w.to(torch.bfloat16).float(), b, 1, (2, 7))[:, 0]
    assert torch.isfinite(out).all()
    e = rel(out, ref)
    assert e < 1e-3, e


@pytest.mark.parametrize("N,H,W,Cx,act,shuffle,res", [(2, 40, 32, 512, 1, 0, False), (3, 1, 16, 1024, 1, 0, False), (3, 1, 16, 256, 0, 0, True),
                                                      (2, 20, 16, 1024, 2, 1, False), (1, 40, 128, 512, 2, 1, False), (2, 1, 128, 5120, 0, 0, False)])
def test_bf16_instnorm_act_matches_fp32(N, H, W, Cx, act, shuffle, res):
    L = lib()
    g = torch.Generator().manual_seed(11)
    x = (1.5 * torch.randn(N, Cx, H, W, generator=g) + 0.7).cuda().to(torch.bfloat16)
    C = Cx // 4 if shuffle else (Cx // 2 if act == 1 else Cx)
    ga, be = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()
    gg, bg = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()
    OH, OW = (2 * H, 2 * W) if shuffle else (H, W)
    r = torch.randn(N, OH, OW, C, generator=g).cuda().to(torch.bfloat16) if res else None
    y = torch.empty(N, OH, OW, C, dtype=torch.bfloat16, device="cuda")
    scratch = torch.zeros(N * 65 * Cx * 2, device="cuda")
    check(L.mcvc_bf16_instnorm_act(ptr(x.permute(0, 2, 3, 1).contiguous()), ptr(ga), ptr(be), ptr(gg), ptr(bg), ptr(r), ptr(y), ptr(scratch),
                                   N, H, W, Cx, act, 1 if shuffle else 0, stream()), "bf16_instnorm_act")
    xf = x.float()
    if shuffle:
        z = F.instance_norm(F.pixel_shuffle(xf, 2), weight=ga, bias=be, eps=1e-5)
    elif act == 1:
        z = F.instance_norm(xf[:, :C], weight=ga, bias=be, eps=1e-5)
        zg = F.instance_norm(xf[:, C:], weight=gg, bias=bg, eps=1e-5)
    else:
        z = F.instance_norm(xf, weight=ga, bias=be, eps=1e-5)
    ref = z * torch.sigmoid(zg) if act == 1 else (z * torch.sigmoid(z) if act == 2 else z)
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    e = rel(y.float().permute(0, 3, 1, 2), ref)
    assert e < 4e-3, e


def _gen(seed):
    g = Generator()
    p = orc.filler_params("G", seed)
    g.load_state_dict(p, strict=True)
    return g.cuda(), p


@pytest.mark.parametrize("B,T", [(1, 64), (2, 68), (1, 65), (3, 32)])
def test_bf16_generator_matches_fp32_oracle(B, T):
    g, p = _gen(21)
    rs = np.random.RandomState(B * 1000 + T)
    x = torch.from_numpy(rs.randn(B, 80, T).astype(np.float32))
    m = torch.from_numpy(orc.fif_mask(rs, B, 80, T, min(25, T)))
    with torch.no_grad():
        ref = orc.generator_forward(p, x, m)
    got = g.infer(x.cuda(), m.cuda(), dtype="bf16")
    assert tuple(got.shape) == tuple(ref.shape)
    e = rel(got.cpu(), ref)
    print("bf16 generator (%d, %d): rel-L2 vs fp32 oracle %.3e" % (B, T, e))
    assert e < 2e-2, e
    # mask=None is the all-ones mask of test.py:92
    ones = g.infer(x.cuda(), None, dtype="bf16")
    ones2 = g.infer(x.cuda(), torch.ones_like(x).cuda(), dtype="bf16")
    assert torch.equal(ones, ones2)


def test_bf16_inference_config_bs16_512_frames():
    """BASELINE configs[4] at full size: bf16 vs the fp32 HIP forward (itself oracle-checked at this size by
    tests/test_hip_model.py) on all 16 samples, and vs the CPU oracle on sample 0; weights = seeded default init cast to bf16."""
    torch.manual_seed(0)
    g = Generator().cuda()
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(16, 80, 512, generator=gen).cuda()
    y32 = g.infer(x, None, dtype="f32")
    y16 = g.infer(x, None, dtype="bf16")
    e = rel(y16, y32)
    print("bf16 vs fp32 HIP at bs=16 x 512: rel-L2 %.3e" % e)
    assert e < 2e-2, e
    per_sample = [rel(y16[i], y32[i]) for i in range(16)]
    assert max(per_sample) < 3e-2, per_sample
    p = {k: v.detach().cpu() for k, v in g.state_dict().items()}
    with torch.no_grad():
        ref0 = orc.generator_forward(p, x[:1].cpu(), torch.ones(1, 80, 512))
    assert rel(y16[:1].cpu(), ref0) < 2e-2


def test_bf16_pack_follows_parameter_updates():
    g, p = _gen(23)
    x = torch.randn(1, 80, 64, device="cuda")
    y0 = g.infer(x, None, dtype="bf16").clone()
    with torch.no_grad():
        for q in g.parameters():
            q.mul_(1.01)                        # in-place update bumps the version counter -> re-pack
    y1 = g.infer(x, None, dtype="bf16")
    assert float((y1 - y0).abs().max()) > 1e-4
    g2 = Generator().cuda()
    g2.load_state_dict(g.state_dict())
    assert torch.equal(g2.infer(x, None, dtype="bf16"), y1)
