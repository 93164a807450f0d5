"""CPU-only checks of the product's host side: module structure, key layout, default init law,
and that the C-ABI library loads and exports every symbol include/mcvc.h declares.
No kernel is launched here (there is no GPU in the build container)."""
import json
import os
import re

import pytest
import torch

from mask_cyclegan_vc import _hip
from mask_cyclegan_vc.model import (Discriminator, DownSampleGenerator, Generator, GLU, PixelShuffle,
                                    ResidualLayer)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mcvc.h")).read()
    declared = set(re.findall(r"\b(mcvc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_hip.EXPORTED_SYMBOLS), declared ^ set(_hip.EXPORTED_SYMBOLS)
    L = _hip.lib()                      # raises if a symbol is missing from the .so
    assert L.mcvc_version() == _hip.ABI_VERSION == 3
    assert L.mcvc_gen_out_frames(64) == 64 and L.mcvc_gen_out_frames(65) == 68   # reference: T=65 -> 68
    assert L.mcvc_disc_out_frames(64) == 8
    assert L.mcvc_gen_packed_floats() > 2 * 24_000_000
    assert L.mcvc_gen_stash_floats(2, 64) == 2 * L.mcvc_gen_stash_floats(1, 64)
    assert L.mcvc_gen_scratch_floats(1, 64) > 0 and L.mcvc_disc_scratch_floats(1, 64) > 0


def test_state_dict_layout_matches_reference(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "layout.json")))
    g, d = Generator(), Discriminator()
    assert [[k, list(v.shape)] for k, v in g.state_dict().items()] == lay["generator_state_dict"]
    assert [n for n, _ in g.named_parameters()] == lay["generator_named_parameters"]
    assert [[k, list(v.shape)] for k, v in d.state_dict().items()] == lay["discriminator_state_dict"]
    assert [n for n, _ in d.named_parameters()] == lay["discriminator_named_parameters"]
    assert [n for n, _ in g.named_children()] == lay["generator_children"]
    assert g.convLayer is g.upSample2
    assert len(list(g.parameters())) == _hip.GEN_NPARAMS and len(list(d.parameters())) == _hip.DISC_NPARAMS
    assert sum(p.numel() for p in g.parameters()) == 24537729
    assert sum(p.numel() for p in d.parameters()) == 16691713
    # strict load of a reference-shaped state dict (114 keys incl. aliases)
    g2 = Generator()
    g2.load_state_dict(g.state_dict(), strict=True)


def test_default_init_is_bit_identical_to_reference_seed0(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "layout.json")))["default_init_seed0"]
    torch.manual_seed(0)
    nets = [Generator(), Generator(), Discriminator(), Discriminator(), Discriminator(), Discriminator()]
    names = ["generator_A2B", "generator_B2A", "discriminator_A", "discriminator_B", "discriminator_A2", "discriminator_B2"]
    for name, net in zip(names, nets):
        ps = list(net.parameters())
        s = float(sum(p.double().abs().sum() for p in ps))
        assert abs(s - lay[name]["abs_sum"]) <= 1e-9 * lay[name]["abs_sum"], name
        assert [float(v) for v in ps[0].flatten()[:4]] == lay[name]["first"]
        assert [float(v) for v in ps[-2].flatten()[:4]] == lay[name]["last"]


def test_no_cpu_fallback():
    g = Generator()
    x = torch.zeros(1, 80, 64)
    with pytest.raises(RuntimeError):
        g(x, torch.ones_like(x))
    with pytest.raises(RuntimeError):
        Discriminator()(x)
    with pytest.raises(RuntimeError):
        GLU()(x)


def test_block_constructors_match_reference_signatures():
    r = ResidualLayer(in_channels=256, out_channels=512, kernel_size=3, stride=1, padding=1)
    assert [k for k in r.state_dict()] == [
        "conv1d_layer.0.weight", "conv1d_layer.0.bias", "conv1d_layer.1.weight", "conv1d_layer.1.bias",
        "conv_layer_gates.0.weight", "conv_layer_gates.0.bias", "conv_layer_gates.1.weight", "conv_layer_gates.1.bias",
        "conv1d_out_layer.0.weight", "conv1d_out_layer.0.bias", "conv1d_out_layer.1.weight", "conv1d_out_layer.1.bias"]
    ds = DownSampleGenerator(in_channels=128, out_channels=256, kernel_size=5, stride=2, padding=2)
    assert ds.convLayer[0].weight.shape == (256, 128, 5, 5)
    assert PixelShuffle(2)(torch.zeros(2, 8, 5)).shape == (2, 4, 10)
    with pytest.raises(ValueError):
        Generator(input_shape=(24, 64))
