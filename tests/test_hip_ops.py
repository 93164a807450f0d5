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
                                      (36, 256, 1024, 96), (3, 128, 64, 16), (2, 256, 200, 48),
                                      (16, 768, 640, 512), (36, 1536, 1400, 64), (12, 640, 3000, 32)])
def test_batched_winograd_gemm_matches_torch(nb, M, N, K):
    """Isolated parity of wino_gemm_kernel (both tile widths) -- the 36 / 16 per-point products of the Winograd convolutions, in the
    shapes the generator produces at B = 1, 2, 4 (incl. the ragged 96- and 160-column cases that take the 32-column tile).  The last three
    shapes take gemm2_kernel's row-group tile order (r6: 8 row tiles per group where a point has >= 8 column tiles) with a row-tile count
    that is NOT a multiple of the group -- 12 tiles of 64 rows (64 x 64 tiles, K >= 512), 12 of 128 rows with a ragged last column tile (the
    128 x 128 tiles of the large grids), 5 of 128 rows (one short group, 12 points) -- so a tile the order skipped or visited twice shows as NaN / a
    wrong block (c is pre-filled with NaN)."""
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


# ---- op-level parity of the kernels that carry the trainer's shapes (r4): Winograd / staged-GEMM layers and the fused trunk backward ----
# (name, Cin, Cout, branches, KH, KW, stride, ph, pw, N, H, W, pixel_shuffle, schemes)
#   schemes: 1 = Winograd, 2x2 output tiles (F(2x2,5x5) / phase F(2x2,3x3)); 2 = 4x4 tiles (F(4x4,5x5) / phase F(4x4,3x3));
#            3 = no Winograd (the discriminators' layers: the direct kernels); 0 = the planner's choice at this shape;
#            5 = implicit GEMMs (forward, data gradient by parity classes, weight gradient with both operands in place)
LAYER_CASES = [
    ("up2.T64", 256, 512, 1, 5, 5, 1, 2, 2, 1, 40, 32, True, (0, 1, 2)),
    ("up2.B2", 256, 512, 1, 5, 5, 1, 2, 2, 2, 40, 32, True, (1, 2)),
    ("up1.T64", 256, 1024, 1, 5, 5, 1, 2, 2, 1, 20, 16, True, (0, 1, 2)),
    ("up1.B3", 256, 1024, 1, 5, 5, 1, 2, 2, 3, 20, 16, True, (1, 2)),
    ("up2.ragged", 256, 512, 1, 5, 5, 1, 2, 2, 1, 18, 10, True, (1,)),             # 45 tiles: the column count is no multiple of 32
    ("up2.T48", 256, 512, 1, 5, 5, 1, 2, 2, 1, 40, 24, True, (1, 2)),
    ("ds1.T64", 128, 256, 2, 5, 5, 2, 2, 2, 1, 80, 64, False, (0, 1, 2)),
    ("ds2.T64", 256, 256, 2, 5, 5, 2, 2, 2, 2, 40, 32, False, (1, 2)),
    ("ds2.B1", 256, 256, 2, 5, 5, 2, 2, 2, 1, 40, 32, False, (1,)),                 # 80 tiles
    ("ds1.ragged", 128, 256, 2, 5, 5, 2, 2, 2, 2, 24, 20, False, (1,)),             # 12 x 10 outputs: 30 tiles per sample
    ("d.ds1", 128, 256, 1, 3, 3, 2, 1, 1, 1, 80, 64, False, (3, 5)),               # 5 = implicit GEMM (forward, data gradient)
    ("d.ds2", 256, 512, 1, 3, 3, 2, 1, 1, 2, 40, 32, False, (3, 5)),
    ("d.ds3", 512, 1024, 1, 3, 3, 2, 1, 1, 3, 20, 16, False, (3, 5)),
    ("d.ds1.T72", 128, 256, 1, 3, 3, 2, 1, 1, 2, 80, 72, False, (3, 5)),
    ("d.ds2.B9", 256, 512, 1, 3, 3, 2, 1, 1, 9, 40, 32, False, (3, 5)),
    ("d.ds3.T128", 512, 1024, 1, 3, 3, 2, 1, 1, 1, 20, 32, False, (5,)),
    ("d.ds1.small", 128, 256, 1, 3, 3, 2, 1, 1, 1, 12, 16, False, (5,)),
    ("d.ds1.B4", 128, 256, 1, 3, 3, 2, 1, 1, 4, 80, 64, False, (5,)),              # K split of the implicit weight gradient over 160 pixel stages
    ("d.ds3.B8", 512, 1024, 1, 3, 3, 2, 1, 1, 8, 20, 16, False, (5,)),             # 128 tiles, no split: straight into the gradient
    ("d.rows12.B9", 256, 768, 1, 3, 3, 2, 1, 1, 9, 40, 32, False, (5,)),           # 12 x 45 forward tiles: igemm's 8-row groups (r6) with a ragged last group of 4
]
_LAYER_IDS = ["%s-s%d" % (c[0], s) for c in LAYER_CASES for s in c[-1]]
_LAYER_PARAMS = [(c, s) for c in LAYER_CASES for s in c[-1]]


@pytest.mark.parametrize("c,scheme", _LAYER_PARAMS, ids=_LAYER_IDS)
def test_layer_ops_match_torch(c, scheme):
    """mcvc_layer_forward / dgrad / wgrad: one convolution layer through the planner the networks use, with the scheme forced -- the
    Winograd transforms (input, output, G^T dU G), the batched products and the staged-GEMM path are reached op by op instead of only
    through whole-network gradients (reference model.py:86-103, 226-237, 298-314) -- against F.conv2d / torch.nn.grad in fp32 on the CPU."""
    from mask_cyclegan_vc._hip import check, lib, ptr, stream
    L = lib()
    name, Cin, Cout, nbr, KH, KW, s, ph, pw, N, H, W, shuffle, _ = c
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g)
    ws = [torch.randn(Cout, Cin, KH, KW, generator=g) / np.sqrt(Cin * KH * KW) for _ in range(nbr)]
    bs = [torch.randn(Cout, generator=g) for _ in range(nbr)]
    wcat, bcat = torch.cat(ws, 0), torch.cat(bs, 0)
    ref = F.conv2d(x, wcat, bcat, stride=s, padding=(ph, pw))
    OH, OW = ref.shape[2], ref.shape[3]
    dy = torch.randn(N, nbr * Cout, OH, OW, generator=g)
    spec = (Cin, Cout, nbr, KH, KW, s, ph, pw)
    packed = torch.zeros(L.mcvc_layer_packed_floats(*spec), device="cuda")
    scratch = torch.zeros(L.mcvc_layer_scratch_floats(N, H, W, *spec), device="cuda")
    wd, bd = [w.cuda() for w in ws], [b.cuda() for b in bs]
    w1, b1 = (wd[1], bd[1]) if nbr == 2 else (None, None)
    check(L.mcvc_layer_pack(ptr(wd[0]), ptr(bd[0]), ptr(w1), ptr(b1), ptr(packed), *spec, stream()), "layer_pack")
    xd, dyd = x.cuda(), dy.cuda()
    # forward (+ the PixelShuffle store of the up-sampling layers)
    y = torch.full((N, nbr * Cout // 4, 2 * OH, 2 * OW) if shuffle else (N, nbr * Cout, OH, OW), float("nan"), device="cuda")
    check(L.mcvc_layer_forward(ptr(xd), ptr(packed), ptr(wd[0]), ptr(w1), ptr(y), ptr(scratch), scratch.numel(), N, H, W, *spec, scheme,
                               1 if shuffle else 0, stream()), "layer_forward")
    want = F.pixel_shuffle(ref, 2) if shuffle else ref
    assert rel_l2(y, want) < 5e-5, rel_l2(y, want)
    # data gradient
    dx = torch.full((N, Cin, H, W), float("nan"), device="cuda")
    check(L.mcvc_layer_dgrad(ptr(dyd), ptr(packed), ptr(wd[0]), ptr(w1), ptr(dx), ptr(scratch), scratch.numel(), N, H, W, *spec, scheme, stream()),
          "layer_dgrad")
    dx_ref = torch.nn.grad.conv2d_input(x.shape, wcat, dy, stride=s, padding=(ph, pw))
    assert rel_l2(dx, dx_ref) < 5e-5, rel_l2(dx, dx_ref)
    # weight gradients (accumulated into zeros; value | gate rows to their own tensors)
    dws = [torch.zeros_like(w) for w in wd]
    check(L.mcvc_layer_wgrad(ptr(xd), ptr(dyd), ptr(dws[0]), ptr(dws[1]) if nbr == 2 else None, ptr(scratch), scratch.numel(), N, H, W, *spec,
                             scheme, stream()), "layer_wgrad")          # (5: the implicit weight gradient, both operands read where they lie)
    dw_ref = torch.nn.grad.conv2d_weight(x, wcat.shape, dy, stride=s, padding=(ph, pw))
    for br in range(nbr):
        e = rel_l2(dws[br], dw_ref[br * Cout:(br + 1) * Cout])
        assert e < 5e-5, (br, e)


def test_layer_ops_refuse_a_winograd_scheme_for_a_layer_without_one():
    from mask_cyclegan_vc._hip import lib, ptr, stream
    L = lib()
    spec = (128, 256, 1, 3, 3, 2, 1, 1)
    packed = torch.zeros(L.mcvc_layer_packed_floats(*spec), device="cuda")
    scratch = torch.zeros(L.mcvc_layer_scratch_floats(1, 16, 16, *spec), device="cuda")
    x, y = torch.zeros(1, 128, 16, 16, device="cuda"), torch.zeros(1, 256, 8, 8, device="cuda")
    w = torch.zeros(256, 128, 3, 3, device="cuda")
    assert L.mcvc_layer_forward(ptr(x), ptr(packed), ptr(w), None, ptr(y), ptr(scratch), scratch.numel(), 1, 16, 16, *spec, 2, 0, stream()) != 0


@pytest.mark.parametrize("B,T4,Cin,Cout,glu", [(1, 16, 256, 512, True), (2, 16, 256, 512, True), (3, 16, 256, 512, True), (4, 16, 256, 512, True),
                                               (1, 16, 512, 256, False), (2, 16, 512, 256, False), (3, 16, 512, 256, False), (4, 16, 512, 256, False),
                                               (3, 12, 256, 512, True), (5, 8, 512, 256, False)])
def test_fused_trunk_layer_backward_matches_autograd(B, T4, Cin, Cout, glu):
    """mcvc_trunk_layer_backward (SURVEY.md section 8b resblock1d_bwd): InstanceNorm (+ gated GLU) backward recomputed inside the transposed
    1x3 convolution launch + the batched small-K weight gradient, against autograd through Conv1d -> InstanceNorm1d -> GLU in fp32
    (reference model.py:47-76) at 16 ... 64 columns."""
    from mask_cyclegan_vc._hip import check, lib, ptr, stream
    L = lib()
    g = torch.Generator().manual_seed(9)
    KW = 3
    x = torch.randn(B, Cin, T4, generator=g, requires_grad=True)
    mk = lambda: (torch.randn(Cout, Cin, KW, generator=g) / (Cin * KW) ** 0.5).requires_grad_()      # noqa: E731
    w, wg = mk(), mk()
    b, bg = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    ga, be = (1 + 0.1 * torch.randn(Cout, generator=g)).requires_grad_(), (0.1 * torch.randn(Cout, generator=g)).requires_grad_()
    gg, bgt = (1 + 0.1 * torch.randn(Cout, generator=g)).requires_grad_(), (0.1 * torch.randn(Cout, generator=g)).requires_grad_()
    c0 = F.conv1d(x, w, b, padding=1)
    c0.retain_grad()
    z = F.instance_norm(c0, weight=ga, bias=be, eps=1e-5)
    if glu:
        c1 = F.conv1d(x, wg, bg, padding=1)
        c1.retain_grad()
        y = z * torch.sigmoid(F.instance_norm(c1, weight=gg, bias=bgt, eps=1e-5))
        conv = torch.cat((c0, c1), 1)
    else:
        y = z
        conv = c0
    dy = torch.randn(B, Cout, T4, generator=g)
    y.backward(dy)
    Cx = conv.shape[1]
    tl = lambda t: t.detach().permute(1, 0, 2).contiguous().cuda()          # noqa: E731  trunk layout [C][B][T4]
    mean = conv.detach().mean(2)
    rstd = 1.0 / torch.sqrt(conv.detach().var(2, unbiased=False) + 1e-5)
    stats = torch.stack((mean, rstd), 2).contiguous().cuda()                 # [B][Cx][2]
    dx = torch.zeros(Cin, B, T4, device="cuda")
    dconv = torch.full((Cx, B, T4), float("nan"), device="cuda")
    z_ = lambda n: torch.zeros(n, device="cuda")                            # noqa: E731
    dga, dbe, dgg, dbg = z_(Cout), z_(Cout), z_(Cout), z_(Cout)
    dw, dwg = torch.zeros(Cout, Cin, KW, device="cuda"), torch.zeros(Cout, Cin, KW, device="cuda")
    wpack = torch.zeros(Cin * Cx * KW, device="cuda")
    wd, wgd = w.detach().cuda(), wg.detach().cuda()
    dy_d, conv_d, x_d = tl(dy), tl(conv), tl(x)                              # (named: a temporary would be freed before the kernel reads it)
    ga_d, be_d, gg_d, bgt_d = ga.detach().cuda(), be.detach().cuda(), gg.detach().cuda(), bgt.detach().cuda()
    check(L.mcvc_trunk_layer_backward(ptr(dy_d), ptr(conv_d), ptr(stats), ptr(ga_d), ptr(be_d),
                                      ptr(gg_d) if glu else None, ptr(bgt_d) if glu else None,
                                      ptr(wd), ptr(wgd) if glu else None, ptr(x_d), ptr(dx), ptr(dconv), ptr(dga), ptr(dbe),
                                      ptr(dgg) if glu else None, ptr(dbg) if glu else None, ptr(dw), ptr(dwg) if glu else None, ptr(wpack),
                                      B, Cin, T4, Cout, KW, stream()), "trunk_layer_backward")
    back = lambda t: t.permute(1, 0, 2)                                      # noqa: E731
    dconv_ref = torch.cat((c0.grad, c1.grad), 1) if glu else c0.grad
    assert rel_l2(back(dconv), dconv_ref) < 5e-5, rel_l2(back(dconv), dconv_ref)
    assert rel_l2(back(dx), x.grad) < 5e-5, rel_l2(back(dx), x.grad)
    assert rel_l2(dga, ga.grad) < 5e-5 and rel_l2(dbe, be.grad) < 5e-5
    assert rel_l2(dw, w.grad) < 5e-5, rel_l2(dw, w.grad)
    if glu:
        assert rel_l2(dgg, gg.grad) < 5e-5 and rel_l2(dbg, bgt.grad) < 5e-5
        assert rel_l2(dwg, wg.grad) < 5e-5, rel_l2(dwg, wg.grad)
