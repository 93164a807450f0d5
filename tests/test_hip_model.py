"""GPU parity of the whole-network HIP path against (a) golden vectors produced by the reference and
(b) the CPU oracle on identical seeded inputs.  Bar: relative L2 <= 1e-3 (north star); observed ~1e-6."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mcvc_oracle as orc  # noqa: E402  (checker only)
from mask_cyclegan_vc.model import Discriminator, DownSampleGenerator, Generator, ResidualLayer  # noqa: E402

TOL = 1e-3


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def seeded_inputs(seed, B, T, max_mask_len=25):
    rs = np.random.RandomState(seed)
    x = rs.randn(B, 80, T).astype(np.float32)
    m = orc.fif_mask(rs, B, 80, T, min(max_mask_len, T))
    return torch.from_numpy(x), torch.from_numpy(m)


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.load(open(os.path.join(golden_dir, "meta.json")))


@pytest.fixture(scope="module")
def nets(meta):
    g = Generator()
    g.load_state_dict(orc.filler_params("G", meta["filler_seeds"]["G"]), strict=True)
    d = Discriminator()
    d.load_state_dict(orc.filler_params("D", meta["filler_seeds"]["D"]), strict=True)
    return g.cuda(), d.cuda()


def test_forward_matches_reference_goldens(golden_dir, meta, nets):
    gold = np.load(os.path.join(golden_dir, "forward.npz"))
    g, d = nets
    worst = 0.0
    with torch.no_grad():
        for case in meta["forward_cases"]:
            B, T = case["B"], case["T"]
            x, m = seeded_inputs(case["seed"], B, T)
            y = g(x.cuda(), m.cuda())
            tag = "%dx%d" % (B, T)
            assert tuple(y.shape) == gold["g_out_" + tag].shape
            e = [rel_l2(y, gold["g_out_" + tag]), rel_l2(d(x.cuda()), gold["d_out_" + tag]), rel_l2(d(y), gold["dg_out_" + tag])]
            worst = max(worst, *e)
            assert max(e) < TOL, (tag, e)
    print("worst forward rel-L2 vs reference goldens: %.3e" % worst)


def test_gradients_match_reference_goldens(golden_dir, meta, nets):
    gold = np.load(os.path.join(golden_dir, "grads.npz"))
    norms = json.load(open(os.path.join(golden_dir, "grad_norms.json")))
    g, d = nets
    c = meta["grad_case"]
    x, m = seeded_inputs(c["seed"], c["B"], c["T"])
    x = x.cuda().requires_grad_(True)
    for p in list(g.parameters()) + list(d.parameters()):
        p.grad = None
    y = g(x, m.cuda())
    loss = torch.mean((1 - d(y)) ** 2) + 10.0 * torch.mean(torch.abs(x.detach() - y))
    loss.backward()
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    assert rel_l2(x.grad, gold["dx"]) < TOL
    checked = 0
    for tag, net in (("G", g), ("D", d)):
        for n, p in net.named_parameters():
            ref = norms[tag + ":" + n]
            if ref is None:
                assert p.grad is None, n                  # dead downSample4: no grad, like the reference
                continue
            if ref < 1e-6:
                # conv bias in front of an InstanceNorm: mathematically zero; the reference carries ~1e-8 noise
                assert float(p.grad.norm()) < 1e-5, n
                continue
            assert abs(float(p.grad.double().norm()) - ref) < TOL * ref, (n, float(p.grad.norm()), ref)
            assert rel_l2(p.grad.flatten()[:32], gold[tag + ":" + n]) < 5e-3, n
            checked += 1
    assert checked > 80


@pytest.mark.parametrize("B,T", [(1, 64), (3, 64), (2, 24), (1, 68), (4, 64), (6, 48)])
def test_full_tensor_parity_vs_oracle(B, T, nets, meta):
    """Every parameter gradient and the input gradient, full tensors, vs the CPU oracle.  With T % 16 == 0, upSample2 (from 4 samples also
    upSample1) runs as F(4x4,5x5) Winograd (64 points, csrc/wino4.h) and from 4 samples downSample1/2 as F(4x4,3x3) over the phase
    planes (36 points, csrc/wino43_kernels.hip) in all three passes; otherwise (T = 24, 68) as F(2x2,5x5) / F(2x2,3x3)."""
    g, d = nets
    gp = orc.filler_params("G", meta["filler_seeds"]["G"])
    dp = orc.filler_params("D", meta["filler_seeds"]["D"])
    x, m = seeded_inputs(77 + B + T, B, T)
    gn, dn = orc.generator_param_names(), orc.discriminator_param_names()
    live_d = [k for k in dn if not k.startswith(orc.DISC_DEAD_PREFIX)]
    xr = x.clone().requires_grad_(True)
    leaves = [xr] + [gp[k].requires_grad_(True) for k in gn] + [dp[k].requires_grad_(True) for k in live_d]
    yr = orc.generator_forward(gp, xr, m)
    dr = orc.discriminator_forward(dp, yr)
    wy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(5))
    loss_r = torch.mean((1 - dr) ** 2) + (yr * wy).mean()
    ref = torch.autograd.grad(loss_r, leaves)

    xd = x.cuda().requires_grad_(True)
    for p in list(g.parameters()) + list(d.parameters()):
        p.grad = None
    y = g(xd, m.cuda())
    dd = d(y)
    assert rel_l2(y, yr) < TOL and rel_l2(dd, dr) < TOL
    loss = torch.mean((1 - dd) ** 2) + (y * wy.cuda()).mean()
    loss.backward()
    assert rel_l2(xd.grad, ref[0]) < TOL
    gd = dict(g.named_parameters()); ddict = dict(d.named_parameters())
    worst = 0.0
    for k, r in zip(gn, ref[1:1 + len(gn)]):
        if float(r.norm()) < 1e-6:
            continue
        e = rel_l2(gd[k].grad, r); worst = max(worst, e)
        assert e < TOL, (k, e)
    for k, r in zip(live_d, ref[1 + len(gn):]):
        if float(r.norm()) < 1e-6:
            continue
        e = rel_l2(ddict[k].grad, r); worst = max(worst, e)
        assert e < TOL, (k, e)
    print("B=%d T=%d worst parameter-gradient rel-L2 vs oracle: %.3e" % (B, T, worst))


def test_building_blocks_match_torch():
    import torch.nn.functional as F
    torch.manual_seed(3)
    r = ResidualLayer(256, 512, 3, 1, 1)
    x = torch.randn(2, 256, 16)
    p = {k: v.detach() for k, v in r.state_dict().items()}

    def inorm(t, n):
        return F.instance_norm(t, None, None, p[n + ".1.weight"], p[n + ".1.bias"], True, 0.0, 1e-5)
    a = inorm(F.conv1d(x, p["conv1d_layer.0.weight"], p["conv1d_layer.0.bias"], 1, 1), "conv1d_layer")
    gt = inorm(F.conv1d(x, p["conv_layer_gates.0.weight"], p["conv_layer_gates.0.bias"], 1, 1), "conv_layer_gates")
    ref = x + inorm(F.conv1d(a * torch.sigmoid(gt), p["conv1d_out_layer.0.weight"], p["conv1d_out_layer.0.bias"], 1, 1), "conv1d_out_layer")
    assert rel_l2(r.cuda()(x.cuda()), ref) < TOL
    ds = DownSampleGenerator(128, 256, 5, 2, 2)
    x2 = torch.randn(1, 128, 20, 16)
    q = {k: v.detach() for k, v in ds.state_dict().items()}
    a = F.instance_norm(F.conv2d(x2, q["convLayer.0.weight"], q["convLayer.0.bias"], 2, 2), None, None, q["convLayer.1.weight"], q["convLayer.1.bias"], True, 0.0, 1e-5)
    gt = F.instance_norm(F.conv2d(x2, q["convLayer_gates.0.weight"], q["convLayer_gates.0.bias"], 2, 2), None, None, q["convLayer_gates.1.weight"], q["convLayer_gates.1.bias"], True, 0.0, 1e-5)
    assert rel_l2(ds.cuda()(x2.cuda()), a * torch.sigmoid(gt)) < TOL


def test_inference_config_bs16_512_frames(nets, meta):
    """BASELINE configs[4] shape (generator_A2B inference, 80 x 512 frames, bs=16; computed in fp32 here, i.e. at or above
    the bf16 the config names): every sample of the batched forward equals its own bs=1 forward (per-sample ops only),
    and sample 0 matches the CPU oracle within 1e-3."""
    g = nets[0]
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(16, 80, 512).astype(np.float32)).cuda()
    mask = torch.ones_like(x)
    with torch.no_grad():
        y = g(x, mask)
        assert y.shape == (16, 80, 512) and bool(torch.isfinite(y).all())
        for i in (0, 7, 15):
            yi = g(x[i:i + 1].contiguous(), mask[i:i + 1].contiguous())
            assert float((y[i:i + 1] - yi).abs().max()) <= 1e-4 * float(yi.abs().max())
    p = {k: v.detach().cpu() for k, v in g.state_dict().items()}
    ref = orc.generator_forward(p, x[:1].cpu(), mask[:1].cpu())
    err = float((y[:1].cpu() - ref).norm() / ref.norm())
    assert err < 1e-3, err


@pytest.mark.parametrize("B,T", [(1, 64), (2, 64), (1, 32), (4, 32), (1, 128), (3, 64), (6, 32), (3, 48)])
def test_persistent_trunk_forward_matches_the_per_layer_launches(B, T):
    """The 12 dependent residual trunk layers (model.py:258-271) run as ONE persistent launch whose workgroups hand activations to each
    other inside the kernel (write-through stores, arrival counter, sc1 loads).  Same products as the per-layer fused kernels; only
    the InstanceNorm statistics are summed in a different order (lane butterfly vs serial) -> outputs AND every stashed intermediate
    agree to 1e-5 of their range; repeated with fresh inputs into the SAME buffers so that a stale L1 / L2 line from the previous pass
    (the hazard of an in-kernel hand-off: a wrong value, not a rounding difference) would show up."""
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    g = Generator()
    g.load_state_dict(orc.filler_params("G", 31), strict=True)
    g = g.cuda()
    ps = list(g.parameters())
    packed = g.packed_weights(ps)
    n_stash, n_scr = L.mcvc_gen_stash_floats(B, T), L.mcvc_gen_scratch_floats(B, T)
    stash = [torch.zeros(n_stash, device="cuda") for _ in range(2)]
    scratch = torch.zeros(n_scr, device="cuda")
    out = [torch.empty(B, 80, L.mcvc_gen_out_frames(T), device="cuda") for _ in range(2)]
    tab = ptr_table(ps)
    was = L.mcvc_set_trunk_persistent(1)
    try:
        for it in range(12):
            x = torch.randn(B, 80, T, device="cuda") * (1.0 + it)
            m = torch.ones_like(x)
            m[:, :, 3 * it:3 * it + 5] = 0
            for k, on in enumerate((1, 0)):
                L.mcvc_set_trunk_persistent(on)
                check(L.mcvc_gen_forward(tab, ptr(packed), ptr(x), ptr(m), ptr(out[k]), ptr(stash[k]), ptr(scratch), n_scr, B, T, stream()), "fwd")
            torch.cuda.synchronize()
            # upSample2 runs as F(4x4,5x5) Winograd, whose fp32 rounding moves by ~1e-5 of the range when its input moves by 1e-6
            # (numpy model in DESIGN.md section 4); a stale line is an O(1) error either way
            tol = 5e-5
            for a_, b_ in ((out[0], out[1]), (stash[0], stash[1])):         # stash: conv outputs, statistics, activations of every layer
                assert float((a_ - b_).abs().max()) <= tol * float(b_.abs().max()), it
    finally:
        L.mcvc_set_trunk_persistent(was)


@pytest.mark.parametrize("B,T", [(1, 64), (1, 32)])
def test_persistent_trunk_backward_matches_the_per_layer_launches(B, T):
    """The backward pass's data-gradient chain through the six residual blocks (12 dependent layers, reference model.py:47-76 under
    autograd) as ONE persistent launch (trunk.h) against one fused launch per layer: the input gradient and EVERY parameter gradient agree
    to rounding (the persistent kernel sums each K range in a fixed order, the per-layer path K-splits with atomics); repeated with fresh
    inputs into the same buffers (a stale line of the in-kernel hand-off would be a wrong value, not a rounding difference); and the
    persistent result is bit-reproducible run to run.  The launch count of the trunk family drops by the 11 saved boundaries."""
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    g = Generator()
    g.load_state_dict(orc.filler_params("G", 37), strict=True)
    g = g.cuda()
    ps = list(g.parameters())
    packed = g.packed_weights(ps, force=True)
    n_stash, n_scr = L.mcvc_gen_stash_floats(B, T), L.mcvc_gen_scratch_floats(B, T)
    stash = torch.zeros(n_stash, device="cuda")
    scratch = torch.zeros(n_scr, device="cuda")
    out = torch.empty(B, 80, L.mcvc_gen_out_frames(T), device="cuda")
    tab = ptr_table(ps)
    grads = [[torch.zeros_like(p) for p in ps] for _ in range(3)]
    gtabs = [ptr_table(gs) for gs in grads]
    dxs = [torch.zeros(B, 80, T, device="cuda") for _ in range(3)]
    was = L.mcvc_set_trunk_persistent(1)
    was_det = L.mcvc_set_deterministic(1)
    counts = {}
    try:
        for it in range(6):
            x = torch.randn(B, 80, T, device="cuda") * (1.0 + it)
            m = torch.ones_like(x)
            m[:, :, 3 * it:3 * it + 5] = 0
            dout = torch.randn(B, 80, T, device="cuda")
            L.mcvc_set_trunk_persistent(1)
            check(L.mcvc_gen_forward(tab, ptr(packed), ptr(x), ptr(m), ptr(out), ptr(stash), ptr(scratch), n_scr, B, T, stream()), "fwd")
            for k, on in enumerate((1, 2, 1)):             # persistent backward, per-layer backward, persistent again
                L.mcvc_set_trunk_persistent(on)
                for gt in grads[k]:
                    gt.zero_()

                def bwd(k=k):
                    check(L.mcvc_gen_backward(tab, ptr(packed), gtabs[k], ptr(m), ptr(dout), ptr(dxs[k]), 0, ptr(stash), ptr(scratch), n_scr, B, T,
                                              stream(), None), "bwd")
                if it == 0 and k < 2:
                    counts[on] = _launches("trunk_layer", bwd)
                else:
                    bwd()
            torch.cuda.synchronize()
            assert float((dxs[0] - dxs[1]).norm()) <= 2e-5 * float(dxs[1].norm()), it
            for i, (a_, b_) in enumerate(zip(grads[0], grads[1])):
                nb = float(b_.norm())
                if nb < 1e-4:                                  # conv biases in front of an InstanceNorm: mathematically zero, rounding noise
                    continue
                assert float((a_ - b_).norm()) <= 2e-5 * nb + 1e-9, (it, i, float((a_ - b_).norm()), nb)
            assert torch.equal(dxs[0], dxs[2])             # (deterministic mode: the layers around the trunk take fixed-order paths too)
            for i in range(len(ps)):
                assert torch.equal(grads[0][i], grads[2][i]), (it, i)
        assert counts[1] == counts[2] - 11, counts
    finally:
        L.mcvc_set_trunk_persistent(was)
        L.mcvc_set_deterministic(was_det)


@pytest.mark.parametrize("B,SB,b0,T", [(2, 3, 0, 64), (1, 2, 0, 64), (1, 3, 0, 64), (1, 3, 1, 64), (1, 3, 2, 64), (2, 3, 1, 64), (2, 3, 0, 32), (2, 4, 2, 32),
                                          (2, 4, 0, 64), (4, 6, 2, 64)])
def test_backward_over_a_window_of_the_forward_samples(B, SB, b0, T):
    """mcvc_gen_backward_window: a forward pass over SB samples, a backward pass through its samples [b0, b0 + B) only (the trainer's merged
    forwards carry the previous iteration's discriminator-phase sample, whose backward the reference discards: train.py:240, 259-273; the
    identity sample's backward runs ahead of the translation sample's).  Every op of the generator is per sample (model.py:239-280), so the
    result must equal forward + backward over those B samples alone: input gradient and every parameter gradient to rounding (the two
    forwards pick different tile shapes / K splits)."""
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    g = Generator()
    g.load_state_dict(orc.filler_params("G", 41), strict=True)
    g = g.cuda()
    ps = list(g.parameters())
    packed = g.packed_weights(ps, force=True)
    tab = ptr_table(ps)
    n_scr = max(L.mcvc_gen_scratch_floats(B, T), L.mcvc_gen_scratch_floats(SB, T))
    scratch = torch.zeros(n_scr, device="cuda")
    stash_s, stash_b = torch.zeros(L.mcvc_gen_stash_floats(SB, T), device="cuda"), torch.zeros(L.mcvc_gen_stash_floats(B, T), device="cuda")
    TO = L.mcvc_gen_out_frames(T)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(SB, 80, T, generator=gen).cuda()
    m = torch.ones_like(x)
    for b in range(SB):
        m[b, :, 4 + 3 * b:9 + 5 * b] = 0
    dout = torch.randn(B, 80, TO, generator=gen).cuda()
    out_s, out_b = torch.empty(SB, 80, TO, device="cuda"), torch.empty(B, 80, TO, device="cuda")
    grads = [[torch.zeros_like(p) for p in ps] for _ in range(2)]
    dxs = [torch.zeros(B, 80, T, device="cuda") for _ in range(2)]
    for persistent in (1, 2, 0):          # persistent forward + backward trunk, forward only, per-layer launches
        was = L.mcvc_set_trunk_persistent(persistent)
        try:
            for gs in grads:
                for gt in gs:
                    gt.zero_()
            check(L.mcvc_gen_forward(tab, ptr(packed), ptr(x), ptr(m), ptr(out_s), ptr(stash_s), ptr(scratch), n_scr, SB, T, stream()), "fwd SB")
            xb, mb = x[b0:b0 + B].contiguous(), m[b0:b0 + B].contiguous()
            check(L.mcvc_gen_backward_window(tab, ptr(packed), ptr_table(grads[0]), ptr(mb), ptr(dout), ptr(dxs[0]), 0, ptr(stash_s), SB, b0, ptr(scratch),
                                             n_scr, B, T, stream(), None, None, 0), "bwd window")
            check(L.mcvc_gen_forward(tab, ptr(packed), ptr(xb), ptr(mb), ptr(out_b), ptr(stash_b), ptr(scratch), n_scr, B, T, stream()), "fwd B")
            check(L.mcvc_gen_backward(tab, ptr(packed), ptr_table(grads[1]), ptr(mb), ptr(dout), ptr(dxs[1]), 0, ptr(stash_b), ptr(scratch), n_scr, B, T,
                                      stream(), None), "bwd B")
            torch.cuda.synchronize()
        finally:
            L.mcvc_set_trunk_persistent(was)
        assert float((out_s[b0:b0 + B] - out_b).norm()) <= 5e-5 * float(out_b.norm()), persistent
        assert float((dxs[0] - dxs[1]).norm()) <= 1e-4 * float(dxs[1].norm()), (persistent, float((dxs[0] - dxs[1]).norm() / dxs[1].norm()))
        for i, (a_, b_) in enumerate(zip(grads[0], grads[1])):
            nb_ = float(b_.norm())
            if nb_ < 1e-4:                                     # conv biases in front of an InstanceNorm: mathematically zero
                continue
            assert float((a_ - b_).norm()) <= 1e-4 * nb_ + 1e-9, (persistent, i, float((a_ - b_).norm()) / nb_)


def _launches(kind_name, fn):
    """Run fn() with the library's per-launch trace on; returns the number of launches of one kernel family."""
    import ctypes
    from mask_cyclegan_vc import _hip
    L = _hip.lib()
    L.mcvc_trace_kind_name.restype = ctypes.c_char_p
    nk = L.mcvc_trace_kinds()
    names = [L.mcvc_trace_kind_name(k).decode() for k in range(nk)]
    buf = (ctypes.c_double * (4 * nk))()
    torch.cuda.synchronize()
    L.mcvc_trace_enable(1)
    try:
        fn()
        L.mcvc_trace_collect(buf)
    finally:
        L.mcvc_trace_enable(0)
    return int(buf[4 * names.index(kind_name)])


@pytest.mark.parametrize("B,T", [(1, 64), (2, 64), (3, 64), (4, 64), (8, 64), (9, 64), (16, 64), (2, 128), (8, 72)])
def test_discriminator_gemm_path_vs_oracle(B, T, nets, meta):
    """The discriminators' stride-2 3x3 layers run as IMPLICIT GEMMs (csrc/sgemm.h): forward and data gradient gather their B operand from
    the phase-split padded input / the padded dY, the data gradient reads the forward weight copy row-major, the weight gradient reads dY and
    the phase-split input in place (wgemm_kernels.hip) -- no tap planes, transposed operands or gather kernel at ANY batch size (r5; r2-r4
    staged them below 4 samples per pass and for every weight gradient).  Whole discriminator, every gradient, vs the CPU oracle
    (model.py:298-349).  B = 1, 2, 3: K-split products whose slabs the consumers sum, K-split weight gradients (slabs + dw_accum); B = 9: pixel
    counts that are no multiple of a tile; B = 16: the large-pass splits; T = 128: other plane shapes; T = 72: planes the implicit kernels do
    not take (36 / 18 / 9 columns) -- the direct kernels."""
    _, d = nets
    dp = orc.filler_params("D", meta["filler_seeds"]["D"])
    dn = orc.discriminator_param_names()
    live_d = [k for k in dn if not k.startswith(orc.DISC_DEAD_PREFIX)]
    x = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(300 + B + T))
    xr = x.clone().requires_grad_(True)
    leaves = [xr] + [dp[k].requires_grad_(True) for k in live_d]
    dr = orc.discriminator_forward(dp, xr)
    w = torch.randn(dr.shape, generator=torch.Generator().manual_seed(6))
    ref = torch.autograd.grad((dr * w).sum(), leaves)
    xd = x.cuda().requires_grad_(True)
    for p in d.parameters():
        p.grad = None
    out = {}

    def run():
        dd = d(xd)
        (dd * w.cuda()).sum().backward()
        out["d"] = dd
    n_gemm = _launches("sgemm", run)
    assert n_gemm == (0 if T == 72 else 9), n_gemm            # 3 layers x {forward, data gradient, weight gradient}; none at T = 72
    assert rel_l2(out["d"], dr) < TOL
    assert rel_l2(xd.grad, ref[0]) < TOL
    ddict = dict(d.named_parameters())
    worst = 0.0
    for k, r in zip(live_d, ref[1:]):
        if k.startswith("downSample") and k.endswith(".0.bias"):      # conv bias in front of an InstanceNorm: its exact gradient is 0,
            assert float(ddict[k].grad.abs().max()) == 0.0            # the oracle's autograd leaves rounding noise there
            continue
        e = rel_l2(ddict[k].grad, r); worst = max(worst, e)
        assert e < TOL, (k, e)
    print("B=%d T=%d implicit-GEMM discriminator: worst parameter-gradient rel-L2 vs oracle %.3e" % (B, T, worst))


def test_generator_repack_in_two_parts_and_its_guard():
    """mcvc_gen_pack_sets: a forward-only refresh (sets = 1) must give the same forward as the full re-pack and must make a backward pass
    on that buffer FAIL (stale data-gradient copies) until the backward sets (sets = 2) have been refreshed; afterwards the gradients
    equal those of the full re-pack bit for bit."""
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    B, T = 1, 64
    g = Generator()
    g.load_state_dict(orc.filler_params("G", 41), strict=True)
    g = g.cuda()
    ps = list(g.parameters())
    tab = ptr_table(ps)
    n_stash, n_scr = L.mcvc_gen_stash_floats(B, T), L.mcvc_gen_scratch_floats(B, T)
    x = torch.randn(B, 80, T, device="cuda"); m = torch.ones_like(x)
    dout = torch.randn(B, 80, T, device="cuda")

    def run(packed, expect_bwd_rc=0):
        stash = torch.zeros(n_stash, device="cuda"); scr = torch.zeros(n_scr, device="cuda")
        out = torch.empty(B, 80, T, device="cuda")
        grads = [torch.zeros_like(p) for p in ps]
        check(L.mcvc_gen_forward(tab, ptr(packed), ptr(x), ptr(m), ptr(out), ptr(stash), ptr(scr), n_scr, B, T, stream()), "fwd")
        rc = L.mcvc_gen_backward(tab, ptr(packed), ptr_table(grads), ptr(m), ptr(dout), None, 0, ptr(stash), ptr(scr), n_scr, B, T, stream(), None)
        torch.cuda.synchronize()
        assert (rc == 0) == (expect_bwd_rc == 0), rc
        return out, grads

    full = torch.zeros(L.mcvc_gen_packed_floats(), device="cuda")
    check(L.mcvc_gen_pack_sets(tab, ptr(full), 2 * B, T, 3, stream()), "pack")
    out_ref, grads_ref = run(full)
    two = torch.zeros_like(full)
    check(L.mcvc_gen_pack_sets(tab, ptr(two), 2 * B, T, 1, stream()), "pack fwd")
    out_fwd, _ = run(two, expect_bwd_rc=1)                      # backward refused: its weight copies were not refreshed
    assert torch.equal(out_fwd, out_ref)
    check(L.mcvc_gen_pack_sets(tab, ptr(two), 2 * B, T, 2, stream()), "pack bwd")
    out2, grads2 = run(two)
    assert torch.equal(out2, out_ref)
    L.mcvc_set_deterministic(1)
    try:
        _, ga = run(full); _, gb = run(two)
        assert all(torch.equal(a, b) for a, b in zip(ga, gb))
    finally:
        L.mcvc_set_deterministic(0)


def test_persistent_trunk_fault_is_reported_not_silent():
    """The persistent trunk kernel's workgroups wait for each other with bounded spins.  If one never arrives (test hook: workgroup 0 skips
    its first arrival) the pass must not hand garbage on: its output is NaN and the scratch buffer's error word names the layer; the
    next healthy pass is bit-identical to the one before the fault."""
    from mask_cyclegan_vc._hip import lib, ptr, stream
    L = lib()
    torch.manual_seed(3)
    g = Generator().cuda()
    x = torch.randn(1, 80, 64, device="cuda")
    y0 = g.infer(x).clone()
    (_stash, scratch), = [v for v in g._iws.values() if isinstance(v, tuple)]
    assert L.mcvc_gen_trunk_fault(ptr(scratch), 1, 64, 1, stream()) >= 0          # (first read clears whatever torch.empty left there)
    assert L.mcvc_gen_trunk_fault(ptr(scratch), 1, 64, 0, stream()) == 0
    assert torch.isfinite(y0).all()
    was = L.mcvc_debug_trunk_fault_inject(1)
    try:
        y1 = g.infer(x).clone()
        torch.cuda.synchronize()
    finally:
        L.mcvc_debug_trunk_fault_inject(was)
    assert torch.isnan(y1).all(), "a lost arrival must poison the pass's result"
    assert L.mcvc_gen_trunk_fault(ptr(scratch), 1, 64, 0, stream()) == 2           # 1 + layer 1: its input (layer 0's rows) never became complete
    assert L.mcvc_gen_trunk_fault(ptr(scratch), 1, 64, 1, stream()) == 2           # sticky until reset ...
    assert L.mcvc_gen_trunk_fault(ptr(scratch), 1, 64, 0, stream()) == 0
    y2 = g.infer(x)
    assert torch.equal(y2, y0) and L.mcvc_gen_trunk_fault(ptr(scratch), 1, 64, 0, stream()) == 0
