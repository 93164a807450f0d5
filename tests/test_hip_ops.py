"""GPU parity of the individual gfx950 kernels (through the C ABI) against plain PyTorch fp32 on CPU.

Tolerance: relative L2 <= 2e-5 for forward/dgrad/wgrad (exact-fp32 MFMA vs mkldnn: summation order
only), InstanceNorm paths <= 5e-5.  The north-star bar is 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mask_cyclegan_vc import ops  # noqa: E402


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


# (Cin, Cout, KH, KW, stride, ph, pw, N, H, W, shuffle)  -- every conv shape of the hot path + ragged sizes
CONV_CASES = [
    ("g.conv1", 2, 256, 5, 15, 1, 2, 7, 2, 80, 64, False),
    ("g.ds1", 128, 512, 5, 5, 2, 2, 2, 1, 80, 64, False),
    ("g.ds2", 256, 512, 5, 5, 2, 2, 2, 2, 40, 32, False),
    ("g.2dto1d", 5120, 256, 1, 1, 1, 0, 0, 1, 2, 16, False),
    ("g.res_vg", 256, 1024, 1, 3, 1, 0, 1, 1, 3, 16, False),
    ("g.res_out", 512, 256, 1, 3, 1, 0, 1, 1, 1, 16, False),
    ("g.1dto2d", 256, 5120, 1, 1, 1, 0, 0, 1, 2, 16, False),
    ("g.up1", 256, 1024, 5, 5, 1, 2, 2, 1, 20, 16, True),
    ("g.up2", 256, 512, 5, 5, 1, 2, 2, 1, 40, 32, True),
    ("g.last", 128, 1, 5, 15, 1, 2, 7, 2, 80, 64, False),
    ("d.conv1", 1, 128, 3, 3, 1, 1, 1, 2, 80, 64, False),
    ("d.ds1", 128, 256, 3, 3, 2, 1, 1, 1, 80, 64, False),
    ("d.ds2", 256, 512, 3, 3, 2, 1, 1, 2, 40, 32, False),
    ("d.ds3", 512, 1024, 3, 3, 2, 1, 1, 2, 20, 16, False),
    ("d.out", 1024, 1, 1, 3, 1, 0, 1, 3, 10, 8, False),
    ("ragged.s2", 6, 40, 5, 5, 2, 2, 2, 2, 13, 21, False),
    ("ragged.s1", 5, 33, 3, 3, 1, 1, 1, 2, 9, 7, False),
    ("ragged.wide", 4, 70, 1, 3, 1, 0, 1, 1, 5, 100, False),
    ("trunk.T4", 256, 1024, 1, 3, 1, 0, 1, 1, 2, 4, False),
    # the 5x15 edge layers on the matrix cores (conv_fewout_mfma_kernel, wgrad_cout1_mfma_kernel): partial channel rounds (6 = 4 + 2
    # channels, 10 = 4 + 4 + 2), a short last row tile (21 = 16 + 5 rows) and a short last band (21 = 8 + 8 + 5)
    ("ragged.last", 6, 1, 5, 15, 1, 2, 7, 3, 21, 64, False),
    ("ragged.conv1", 2, 10, 5, 15, 1, 2, 7, 2, 21, 64, False),
]


def _case(c, seed):
    name, Cin, Cout, KH, KW, s, ph, pw, N, H, W, sh = c
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / np.sqrt(Cin * KH * KW)
    b = torch.randn(Cout, generator=g)
    return x, w, b


@pytest.mark.parametrize("c", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward(c):
    name, Cin, Cout, KH, KW, s, ph, pw, N, H, W, sh = c
    x, w, b = _case(c, 1)
    ref = F.conv2d(x, w, b, s, (ph, pw))
    if sh:
        ref = F.pixel_shuffle(ref, 2)
    y = ops.conv2d_forward(x.cuda(), w.cuda(), b.cuda(), s, (ph, pw), sh)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < 2e-5, name


@pytest.mark.parametrize("c", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_dgrad(c):
    name, Cin, Cout, KH, KW, s, ph, pw, N, H, W, sh = c
    x, w, b = _case(c, 2)
    OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
    dy = torch.randn(N, Cout, OH, OW, generator=torch.Generator().manual_seed(3))
    ref = torch.nn.grad.conv2d_input(x.shape, w, dy, s, (ph, pw))
    dx = ops.conv2d_dgrad(dy.cuda(), w.cuda(), tuple(x.shape), s, (ph, pw))
    assert rel_l2(dx, ref) < 2e-5, name


@pytest.mark.parametrize("c", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_wgrad(c):
    name, Cin, Cout, KH, KW, s, ph, pw, N, H, W, sh = c
    x, w, b = _case(c, 4)
    OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
    dy = torch.randn(N, Cout, OH, OW, generator=torch.Generator().manual_seed(5))
    ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, s, (ph, pw))
    dw = ops.conv2d_wgrad(x.cuda(), dy.cuda(), tuple(w.shape), s, (ph, pw))
    assert rel_l2(dw, ref) < 2e-5, name
    db = ops.bias_grad(dy.cuda())
    assert rel_l2(db, dy.sum((0, 2, 3))) < 2e-5


NORM_CASES = [  # (N, C, H, W, act)
    (2, 256, 40, 32, ops.ACT_GLU), (1, 256, 20, 16, ops.ACT_GLU), (3, 512, 1, 16, ops.ACT_GLU),
    (2, 256, 1, 16, ops.ACT_NONE), (1, 5120, 1, 16, ops.ACT_NONE), (2, 128, 80, 64, ops.ACT_SILU),
    (2, 1024, 10, 8, ops.ACT_SILU), (1, 256, 40, 32, ops.ACT_SILU), (2, 7, 3, 5, ops.ACT_GLU), (1, 512, 1, 4, ops.ACT_GLU),
]


def _norm_ref(x, gamma, beta, gamma_g, beta_g, res, act, C):
    if act == ops.ACT_GLU:
        a = F.instance_norm(x[:, :C], None, None, gamma, beta, True, 0.0, 1e-5)
        g = F.instance_norm(x[:, C:], None, None, gamma_g, beta_g, True, 0.0, 1e-5)
        y = a * torch.sigmoid(g)
    else:
        z = F.instance_norm(x, None, None, gamma, beta, True, 0.0, 1e-5)
        y = z * torch.sigmoid(z) if act == ops.ACT_SILU else z
    return y if res is None else y + res


@pytest.mark.parametrize("c", NORM_CASES, ids=["%dx%dx%dx%d-a%d" % c for c in NORM_CASES])
def test_instnorm_act_forward_backward(c):
    N, C, H, W, act = c
    g = torch.Generator().manual_seed(11)
    Cx = 2 * C if act == ops.ACT_GLU else C
    x = (torch.randn(N, Cx, H, W, generator=g) * 1.7 + 0.3).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    gg = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True) if act == ops.ACT_GLU else None
    bg = (0.3 * torch.randn(C, generator=g)).requires_grad_(True) if act == ops.ACT_GLU else None
    res = torch.randn(N, C, H, W, generator=g).requires_grad_(True) if act == ops.ACT_NONE else None
    dy = torch.randn(N, C, H, W, generator=g)
    ref = _norm_ref(x, gamma, beta, gg, bg, res, act, C)
    leaves = [t for t in (x, gamma, beta, gg, bg, res) if t is not None]
    ref_grads = torch.autograd.grad(ref, leaves, dy)

    dev = [t.detach().cuda().requires_grad_(True) if t is not None else None for t in (x, gamma, beta, gg, bg, res)]
    y = ops.instnorm_act(dev[0], dev[1], dev[2], act, dev[3], dev[4], dev[5])
    assert rel_l2(y, ref) < 5e-5
    grads = torch.autograd.grad(y, [t for t in dev if t is not None], dy.cuda())
    for gr, rg in zip(grads, ref_grads):
        assert rel_l2(gr, rg) < 2e-4


@pytest.mark.parametrize("act", [ops.ACT_GLU, ops.ACT_SILU, ops.ACT_SIGMOID])
def test_activation_forward_backward(act):
    g = torch.Generator().manual_seed(13)
    C = 6
    x = torch.randn(2, 2 * C if act == ops.ACT_GLU else C, 5, 7, generator=g).requires_grad_(True)
    if act == ops.ACT_GLU:
        ref = x[:, :C] * torch.sigmoid(x[:, C:])
    elif act == ops.ACT_SILU:
        ref = x * torch.sigmoid(x)
    else:
        ref = torch.sigmoid(x)
    dy = torch.randn(ref.shape, generator=g)
    (rg,) = torch.autograd.grad(ref, x, dy)
    xd = x.detach().cuda().requires_grad_(True)
    y = ops.activation(xd, act)
    (gd,) = torch.autograd.grad(y, xd, dy.cuda())
    assert rel_l2(y, ref) < 1e-6 and rel_l2(gd, rg) < 1e-6


def test_losses_and_adam():
    from mask_cyclegan_vc import _hip
    from mask_cyclegan_vc._hip import check, lib, ptr, stream
    L = lib()
    g = torch.Generator().manual_seed(17)
    a = torch.randn(3, 80, 64, generator=g); b = torch.randn(3, 80, 64, generator=g)
    slots = torch.zeros(4, device="cuda")
    ga = torch.empty_like(a, device="cuda")
    ad, bd = a.cuda(), b.cuda()
    check(L.mcvc_l1_loss(ptr(ad), ptr(bd), a.numel(), 10.0, ptr(slots[0:1]), ptr(slots[1:2]), ptr(ga), 0, stream()))
    ar = a.clone().requires_grad_(True)
    l1 = torch.mean(torch.abs(ar - b))
    (gr,) = torch.autograd.grad(10.0 * l1, ar)
    assert abs(float(slots[1]) - float(l1)) < 1e-6 and abs(float(slots[0]) - 10 * float(l1)) < 1e-5
    assert rel_l2(ga, gr) < 1e-6
    # LSGAN on sigmoid outputs, gradient w.r.t. the logits
    z = torch.randn(2, 1, 10, 8, generator=g)
    zr = z.clone().requires_grad_(True)
    for target in (1.0, 0.0):
        d = torch.sigmoid(zr)
        loss = 0.5 * torch.mean((target - d) ** 2)
        (gz,) = torch.autograd.grad(loss, zr)
        slots.zero_()
        gl = torch.empty(z.shape, device="cuda")
        dd = torch.sigmoid(z).cuda()
        check(L.mcvc_lsgan_loss(ptr(dd), z.numel(), target, 0.5, ptr(slots[0:1]), ptr(slots[1:2]), ptr(gl), stream()))
        assert abs(float(slots[0]) - float(loss)) < 1e-6
        assert rel_l2(gl, gz) < 1e-5
    # Adam vs torch.optim.Adam over 3 steps
    n = 1003 * 4 + 3
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.5, 0.999))
    pd = torch.zeros(n + 1, device="cuda")[:n]   # 16-byte aligned base
    pd.copy_(p0)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 0.01
        pr.grad = gr.clone()
        opt.step()
        gd = gr.cuda()
        check(L.mcvc_adam_step(ptr(pd), ptr(gd), ptr(m), ptr(v), n, 2e-4, 0.5, 0.999, 1e-8, step, 1.0, stream()))
    assert float((pd.cpu() - pr.detach()).abs().max()) < 2e-7


@pytest.mark.parametrize("B,T4,Cin,Cout,KW,kind", [(1, 16, 256, 512, 3, "glu"), (2, 16, 256, 512, 3, "glu"), (2, 16, 512, 256, 3, "res"),
                                                   (1, 16, 512, 256, 3, "res"), (2, 16, 256, 5120, 1, "plain"), (4, 8, 256, 512, 3, "glu"),
                                                   (1, 32, 512, 256, 3, "res"),
                                                   # r4: up to 64 columns (three / four 16-column accumulators per wave) and ragged column counts
                                                   (3, 16, 256, 512, 3, "glu"), (3, 16, 512, 256, 3, "res"), (3, 16, 256, 5120, 1, "plain"),
                                                   (4, 16, 256, 512, 3, "glu"), (5, 8, 256, 512, 3, "glu"),
                                                   (3, 12, 512, 256, 3, "res"), (2, 20, 256, 5120, 1, "plain")])
def test_fused_trunk_layer_matches_torch(B, T4, Cin, Cout, KW, kind):
    """Isolated parity of trunk_layer_kernel (SURVEY.md section 8b resblock1d / gemm1x1_in): Conv1d + InstanceNorm1d + gated GLU /
    residual in one launch vs the same ops in plain PyTorch fp32 (reference model.py:47-76, 266-267)."""
    import torch.nn.functional as F
    from mask_cyclegan_vc._hip import check, lib, ptr, stream
    L = lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, T4, generator=g).cuda()
    mk = lambda *s: (torch.randn(*s, generator=g) / (Cin * KW) ** 0.5).cuda()      # noqa: E731
    w, b = mk(Cout, Cin, KW), torch.randn(Cout, generator=g).cuda()
    ga, be = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda(), (0.1 * torch.randn(Cout, generator=g)).cuda()
    wg, bg = mk(Cout, Cin, KW), torch.randn(Cout, generator=g).cuda()
    gg, bgt = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda(), (0.1 * torch.randn(Cout, generator=g)).cuda()
    res = torch.randn(B, Cout, T4, generator=g).cuda()
    xt = x.permute(1, 0, 2).contiguous()                     # trunk layout [C][B][T4]
    rt = res.permute(1, 0, 2).contiguous()
    ncx = 2 * Cout if kind == "glu" else Cout
    conv_out = torch.empty(ncx, B, T4, device="cuda"); stats = torch.empty(B, ncx, 2, device="cuda"); y = torch.empty(Cout, B, T4, device="cuda")
    glu = kind == "glu"
    check(L.mcvc_trunk_layer_forward(ptr(xt), ptr(w), ptr(b), ptr(ga), ptr(be), ptr(wg) if glu else None, ptr(bg) if glu else None,
                                     ptr(gg) if glu else None, ptr(bgt) if glu else None, ptr(rt) if kind == "res" else None, ptr(conv_out),
                                     ptr(stats), ptr(y), B, Cin, T4, Cout, KW, stream()), "trunk_layer_forward")
    pad = (KW - 1) // 2
    c0 = F.conv1d(x, w, b, padding=pad)
    z = F.instance_norm(c0, weight=ga, bias=be, eps=1e-5)
    if glu:
        c1 = F.conv1d(x, wg, bg, padding=pad)
        ref = z * torch.sigmoid(F.instance_norm(c1, weight=gg, bias=bgt, eps=1e-5))
        ref_conv = torch.cat((c0, c1), 1)
    else:
        ref = z + res if kind == "res" else z
        ref_conv = c0
    got = y.permute(1, 0, 2)
    assert float((got - ref).norm() / ref.norm()) < 2e-5
    assert float((conv_out.permute(1, 0, 2) - ref_conv).norm() / ref_conv.norm()) < 2e-5
    mean = ref_conv.mean(2)
    assert float((stats[:, :, 0] - mean).abs().max()) < 1e-4


@pytest.mark.parametrize("nb,M,N,K", [(36, 1024, 96, 256), (36, 1024, 160, 256), (36, 512, 320, 256), (16, 512, 96, 1024), (16, 512, 640, 512),
                                      (36, 256, 1024, 96), (3, 128, 64, 16), (2, 256, 200, 48)])
def test_batched_winograd_gemm_matches_torch(nb, M, N, K):
    """Isolated parity of wino_gemm_kernel (both tile widths) -- the 36 / 16 per-point products of the Winograd convolutions, in the
    shapes the generator produces at B = 1, 2, 4 (incl. the ragged 96- and 160-column cases that take the 32-column tile)."""
    from mask_cyclegan_vc._hip import check, lib, ptr, stream
    L = lib()
    g = torch.Generator().manual_seed(3)
    ldb = (N + 31) // 32 * 32
    if ldb < 64:
        ldb = 64
    a = torch.randn(nb, K, M, generator=g).cuda()
    b = torch.zeros(nb, K, ldb, device="cuda"); b[:, :, :N] = torch.randn(nb, K, N, generator=g).cuda()
    c = torch.full((nb, M, ldb), float("nan"), device="cuda")
    check(L.mcvc_batched_gemm(ptr(a), ptr(b), ptr(c), nb, M, N, K, M, ldb, ldb, K * M, K * ldb, M * ldb, stream()), "batched_gemm")
    ref = torch.bmm(a.transpose(1, 2).double(), b[:, :, :N].double())
    got = c[:, :, :N].double()
    assert torch.isfinite(got).all()
    assert float((got - ref).norm() / ref.norm()) < 2e-6
