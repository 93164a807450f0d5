"""optimizer.step() fused with the weight re-pack (mcvc_gen_update_ranges / mcvc_disc_update_batch, reference train.py:242 / :299).

The fused launch must leave EXACTLY what the two-launch form (mcvc_adam_step2 on the flat range, then the re-pack of that range) leaves:
parameters, both moments, the cleared gradient(s) and every float of the packed buffer, bit for bit -- at the per-pass batches of the
bench configurations (1, 8, 32 samples per GPU: they select different sets of packed copies) and for each parameter range on its own."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import mcvc_oracle as orc  # noqa: E402
from mask_cyclegan_vc.model import Discriminator, Generator  # noqa: E402


def _flat_group(mod, live=None):
    """The engine's layout (engine._FlatGroup): every live parameter at a 4-float-aligned offset of ONE flat buffer; gradients and moments at
    the same offsets of theirs."""
    ps = list(mod.parameters())
    live = list(range(len(ps))) if live is None else live
    offs, off = {}, 0
    for i in live:
        offs[i] = off
        off += (ps[i].numel() + 3) // 4 * 4
    flat = torch.zeros(off, device="cuda")
    for i in live:
        v = flat[offs[i]:offs[i] + ps[i].numel()].view(ps[i].shape)
        v.copy_(ps[i].data)
        ps[i].data = v
    return ps, flat, offs


def _state(n, seed, ps, offs):
    """Random gradients / moments; zero on the alignment padding between tensors, as in the engine (nothing ever writes there: the flat Adam
    launch leaves those elements at 0, the fused update does not visit them)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    valid = torch.zeros(n, device="cuda")
    for i, o in offs.items():
        valid[o:o + ps[i].numel()] = 1.0
    grad = torch.randn(n, device="cuda", generator=g) * 1e-2 * valid
    grad2 = torch.randn(n, device="cuda", generator=g) * 1e-2 * valid
    m = torch.randn(n, device="cuda", generator=g) * 1e-3 * valid
    v = torch.rand(n, device="cuda", generator=g) * 1e-4 * valid
    return grad, grad2, m, v


HYPER = dict(lr=2e-4, b1=0.5, b2=0.999, eps=1e-8, step=7, scale=0.5)


@pytest.mark.parametrize("use_g2", [False, True])
@pytest.mark.parametrize("max_batch", [2, 16, 64])
def test_generator_update_equals_adam_then_repack(max_batch, use_g2):
    import ctypes
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    T = 64
    g = Generator()
    g.load_state_dict(orc.filler_params("G", 11), strict=True)
    g = g.cuda()
    ps, flat, offs = _flat_group(g)
    tab = ptr_table(ps)
    numel = (ctypes.c_longlong * len(ps))(*[p.numel() for p in ps])
    n = flat.numel()
    grad, grad2, m, v = _state(n, 5, ps, offs)
    h = HYPER
    # parameter ranges of the generator: [100,110), [24,100), [0,24) = range_mask bits 0, 1, 2
    # ... and the head in three parts, [12,24), [4,12), [0,4) = bits 3, 4, 5 (what the engine's default single-GPU update issues behind the
    # MCVC_BWD_FINE_MILESTONES events: engine._g_parts.update_range)
    bounds = {1: (offs[100], n), 2: (offs[24], offs[100]), 4: (0, offs[24]), 8: (offs[12], offs[24]), 16: (offs[4], offs[12]), 32: (0, offs[4])}

    def two_launches(mask_list):
        keep = flat.clone()
        gr, g2, mm, vv = grad.clone(), grad2.clone(), m.clone(), v.clone()
        packed = torch.zeros(L.mcvc_gen_packed_floats(), device="cuda")
        for mask in mask_list:
            for bit in (1, 2, 4, 8, 16, 32):
                if mask & bit:
                    lo, hi = bounds[bit]
                    check(L.mcvc_adam_step2(ptr(flat[lo:hi]), ptr(gr[lo:hi]), ptr(g2[lo:hi]) if use_g2 else None, 1, ptr(mm[lo:hi]), ptr(vv[lo:hi]),
                                            hi - lo, h["lr"], h["b1"], h["b2"], h["eps"], h["step"], h["scale"], stream()), "adam")
            check(L.mcvc_gen_pack_ranges(tab, ptr(packed), max_batch, T, 3, mask, stream()), "pack")
        torch.cuda.synchronize()
        f = flat.clone()
        flat.copy_(keep)
        return f, gr, g2, mm, vv, packed

    def fused(mask_list):
        keep = flat.clone()
        gr, g2, mm, vv = grad.clone(), grad2.clone(), m.clone(), v.clone()
        packed = torch.zeros(L.mcvc_gen_packed_floats(), device="cuda")
        for mask in mask_list:
            check(L.mcvc_gen_update_ranges(tab, numel, ptr(packed), max_batch, T, mask, ptr(flat), ptr(gr), ptr(g2) if use_g2 else None, ptr(mm), ptr(vv),
                                           h["lr"], h["b1"], h["b2"], h["eps"], h["step"], h["scale"], 1, stream()), "update")
        torch.cuda.synchronize()
        f = flat.clone()
        flat.copy_(keep)
        return f, gr, g2, mm, vv, packed

    names = ("parameters", "grad", "grad2", "exp_avg", "exp_avg_sq", "packed")
    whole = None
    # all ranges in one launch; range by range (the engine's ranged update); the head in three parts (the DEFAULT single-GPU schedule:
    # masks 1, 2, 8, 16, 32 in the order a backward pass finishes them); the three parts in one launch
    for masks in ([7], [1, 2, 4], [1, 2, 8, 16, 32], [3, 56]):
        ref, got = two_launches(masks), fused(masks)
        for name, a, b in zip(names, ref, got):
            assert torch.equal(a, b), (masks, name, int((a != b).sum()), float((a - b).abs().max()))
        if whole is None:
            whole = got
        for name, a, b in zip(names, whole, got):          # every partition of the parameters leaves the same state as the single launch
            assert torch.equal(a, b), (masks, "vs [7]", name, int((a != b).sum()))
        assert float(got[1].abs().max()) == 0.0                                   # gradients cleared behind the read
        assert float(got[2].abs().max()) == (0.0 if use_g2 else float(grad2.abs().max()))
        assert not torch.equal(got[0], flat)                                       # (and something was updated)


    # the head alone: its three parts == the head as one range, and nothing outside [0, 24) is touched
    a, b = fused([8, 16, 32]), fused([4])
    for name, x, y in zip(names, a, b):
        assert torch.equal(x, y), ("[8,16,32] vs [4]", name, int((x != y).sum()))
    assert torch.equal(a[0][offs[24]:], flat[offs[24]:]) and torch.equal(a[1][offs[24]:], grad[offs[24]:])
    # bit 2 is the UNION of bits 3-5: a mask naming both would update a tensor twice and is refused, by the update and by the re-pack
    packed = torch.zeros(L.mcvc_gen_packed_floats(), device="cuda")
    gr, mm, vv = grad.clone(), m.clone(), v.clone()
    for bad in (4 | 8, 4 | 16, 4 | 32, 7 | 56, 0, 64):
        assert L.mcvc_gen_update_ranges(tab, numel, ptr(packed), max_batch, T, bad, ptr(flat), ptr(gr), None, ptr(mm), ptr(vv),
                                        h["lr"], h["b1"], h["b2"], h["eps"], h["step"], h["scale"], 1, stream()) != 0, bad
        assert L.mcvc_gen_pack_ranges(tab, ptr(packed), max_batch, T, 3, bad, stream()) != 0, bad
    torch.cuda.synchronize()
    assert torch.equal(gr, grad) and torch.equal(mm, m)        # (a refused call did nothing)


@pytest.mark.parametrize("persistent", [1, 0])
@pytest.mark.parametrize("B", [1, 2])
def test_backward_milestones_fire_after_their_gradient_ranges_are_final(B, persistent):
    """the default single-GPU generator update runs on another lane BESIDE the last backward pass, range by range, each behind
    one of the pass's four milestone events (include/mcvc.h: [0] parameters [100,110), [1] [24,100), MCVC_BWD_FINE_MILESTONES: [2] [12,24),
    [3] [4,12)); the update rewrites weights and clears gradients in place, so an event that fires before its range's last weight-gradient
    kernel (main or auxiliary stream) is a race.  Here a second stream waits for each event and SNAPSHOTS the range at that moment; every
    snapshot must equal the gradients after the whole pass, bit for bit (nothing may add to a range once its event has fired).  The
    gradient buffers start from a non-zero fill, so a kernel that has not yet run shows as a difference, never as 0 == 0."""
    import ctypes
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    T = 64
    g = Generator()
    g.load_state_dict(orc.filler_params("G", 17), strict=True)
    g = g.cuda()
    ps, flat, offs = _flat_group(g)
    n = flat.numel()
    packed = g.packed_weights(ps, force=True)
    tab = ptr_table(ps)
    grad = torch.zeros(n, device="cuda")
    gviews = [grad[offs[i]:offs[i] + p.numel()].view(p.shape) for i, p in enumerate(ps)]
    gtab = ptr_table(gviews)
    n_scr = L.mcvc_gen_scratch_floats(B, T)
    scratch, stash = torch.zeros(n_scr, device="cuda"), torch.zeros(L.mcvc_gen_stash_floats(B, T), device="cuda")
    L.mcvc_gen_trunk_fault(ptr(scratch), B, T, 1, stream())
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, 80, T, generator=gen).cuda()
    msk = torch.ones_like(x)
    msk[:, :, 10:21] = 0
    dout = torch.randn(B, 80, T, generator=gen).cuda()
    out = torch.empty(B, 80, T, device="cuda")
    ranges = [(offs[100], n), (offs[24], offs[100]), (offs[12], offs[24]), (offs[4], offs[12])]
    evs = [torch.cuda.Event() for _ in range(4)]
    for e in evs:
        e.record()
    ms = (ctypes.c_void_p * 4)(*[e.cuda_event for e in evs])
    aux, side = torch.cuda.Stream(), torch.cuda.Stream()
    snaps = [torch.empty(hi - lo, device="cuda") for lo, hi in ranges]
    was = L.mcvc_set_trunk_persistent(persistent)
    try:
        for rep in range(6):
            grad.fill_(0.125 * (rep + 1))
            check(L.mcvc_gen_forward(tab, ptr(packed), ptr(x), ptr(msk), ptr(out), ptr(stash), ptr(scratch), n_scr, B, T, stream()), "fwd")
            torch.cuda.synchronize()
            check(L.mcvc_gen_backward_window(tab, ptr(packed), gtab, ptr(msk), ptr(dout), None, 0, ptr(stash), B, 0, ptr(scratch), n_scr, B, T,
                                             stream(), ctypes.c_void_p(aux.cuda_stream), ms, 2), "bwd")
            with torch.cuda.stream(side):
                for k, (lo, hi) in enumerate(ranges):
                    side.wait_event(evs[k])
                    snaps[k].copy_(grad[lo:hi])
            torch.cuda.synchronize()
            for k, (lo, hi) in enumerate(ranges):
                assert torch.equal(snaps[k], grad[lo:hi]), (rep, k, int((snaps[k] != grad[lo:hi]).sum()))
            live = [i for i, gv in enumerate(gviews) if float((gv - 0.125 * (rep + 1)).abs().max()) > 0]
            assert len(live) >= 80                              # (the pass did write: all but the zero-gradient biases moved off the fill)
    finally:
        L.mcvc_set_trunk_persistent(was)


@pytest.mark.parametrize("max_batch", [2, 16, 64])
def test_discriminator_update_equals_adam_then_repack(max_batch):
    import ctypes
    from mask_cyclegan_vc._hip import check, lib, ptr, ptr_table, stream
    L = lib()
    T = 64
    d = Discriminator()
    d.load_state_dict(orc.filler_params("D", 12), strict=True)
    d = d.cuda()
    names = [k for k, _ in d.named_parameters()]
    live = [i for i, k in enumerate(names) if "downSample4" not in k]
    assert len(live) < len(names)
    ps, flat, offs = _flat_group(d, live)
    tab = ptr_table(ps)
    numel = (ctypes.c_longlong * len(ps))(*[(p.numel() if i in offs else 0) for i, p in enumerate(ps)])
    n = flat.numel()
    grad, _, m, v = _state(n, 6, ps, offs)
    h = HYPER
    keep = flat.clone()
    # two launches
    gr, mm, vv = grad.clone(), m.clone(), v.clone()
    packed_ref = torch.zeros(L.mcvc_disc_packed_floats(), device="cuda")
    check(L.mcvc_adam_step2(ptr(flat), ptr(gr), None, 1, ptr(mm), ptr(vv), n, h["lr"], h["b1"], h["b2"], h["eps"], h["step"], h["scale"], stream()), "adam")
    check(L.mcvc_disc_pack_batch(tab, ptr(packed_ref), max_batch, T, stream()), "pack")
    torch.cuda.synchronize()
    ref = (flat.clone(), gr, mm, vv, packed_ref)
    flat.copy_(keep)
    # fused
    gr2, mm2, vv2 = grad.clone(), m.clone(), v.clone()
    packed = torch.zeros_like(packed_ref)
    check(L.mcvc_disc_update_batch(tab, numel, ptr(packed), max_batch, T, ptr(flat), ptr(gr2), None, ptr(mm2), ptr(vv2),
                                   h["lr"], h["b1"], h["b2"], h["eps"], h["step"], h["scale"], 1, stream()), "update")
    torch.cuda.synchronize()
    got = (flat.clone(), gr2, mm2, vv2, packed)
    for name, a, b in zip(("parameters", "grad", "exp_avg", "exp_avg_sq", "packed"), ref, got):
        assert torch.equal(a, b), (name, int((a != b).sum()), float((a - b).abs().max()))
    assert not torch.equal(got[0], keep)


def test_update_rejects_a_wrong_element_count():
    """The owner tiles are built from the planner's layer geometry; a parameter table that does not match it must be refused, not updated."""
    import ctypes
    from mask_cyclegan_vc._hip import lib, ptr, ptr_table, stream
    L = lib()
    g = Generator().cuda()
    ps, flat, _ = _flat_group(g)
    numel = (ctypes.c_longlong * len(ps))(*[p.numel() for p in ps])
    numel[104] += 4                                  # upSample1's weight
    z = torch.zeros_like(flat)
    packed = torch.zeros(L.mcvc_gen_packed_floats(), device="cuda")
    rc = L.mcvc_gen_update_ranges(ptr_table(ps), numel, ptr(packed), 2, 64, 1, ptr(flat), ptr(z), None, ptr(z.clone()), ptr(z.clone()),
                                  2e-4, 0.5, 0.999, 1e-8, 1, 1.0, 1, stream())
    assert rc != 0
