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
