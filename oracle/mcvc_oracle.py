"""CPU ORACLE for the MaskCycleGAN-VC training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, pure-PyTorch *CPU* restatement of the reference's
arithmetic for the one hot path this repo accelerates (Generator / Discriminator forward and the
G+D training step).  It is the checker, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
  * the product (``maskcyclegan-vc_amd/``) never imports anything from ``oracle/`` and fails loudly
    when the HIP library is missing -- there is no CPU fallback in the product.

Parity pin: the reference holds no tests or golden vectors of its own, so this
oracle is pinned against outputs of the *reference itself*, imported from /root/reference in the
build container by ``tests/golden/make_golden.py`` and frozen as small fixtures under
``tests/golden/*.npz|json`` (``tests/test_oracle_golden.py`` replays them).  The arithmetic lives in
third-party PyTorch ATen (F.conv2d / F.conv1d / F.instance_norm / F.pixel_shuffle / sigmoid), the
same library the reference calls; the restatement below is about *structure* (which op, in which
order, with which quirk), each function citing the reference file:line it follows
(paths relative to the reference repo root).

Weights are held in plain ``dict[str, Tensor]`` keyed by the reference's ``state_dict`` names
(SURVEY.md Appendix B), so fixtures, the oracle and the HIP-backed modules share one vocabulary.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

IN_EPS = 1e-5  # nn.InstanceNorm{1,2}d default eps; biased variance; no running stats
NORM_KEYS: set = set()     # generator InstanceNorm affine keys (filled by generator_key_shapes)
D_NORM_KEYS: set = set()   # discriminator InstanceNorm affine keys

# ----------------------------------------------------------------------------------------------
# state_dict layouts (reference: mask_cyclegan_vc/model.py:110-211 for G, :287-327 for D)
# ----------------------------------------------------------------------------------------------


def generator_key_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    """The 114 state_dict entries of ``Generator()`` in registration order.

    ``convLayer.*`` is the alias the reference creates by assigning ``self.convLayer`` inside
    ``upsample()`` (model.py:226-237): after construction it is the same module object as
    ``upSample2`` and is registered *before* ``upSample1`` in ``_modules`` order.
    """
    ks: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, shape):
        ks[name + ".weight"] = tuple(shape)
        ks[name + ".bias"] = (shape[0],)

    def norm(name, c):
        ks[name + ".weight"] = (c,)
        ks[name + ".bias"] = (c,)
        NORM_KEYS.update((name + ".weight", name + ".bias"))

    conv("conv1", (128, 2, 5, 15))                       # model.py:116-120
    conv("conv1_gates", (128, 2, 5, 15))                 # model.py:122-126
    for ds, cin in (("downSample1", 128), ("downSample2", 256)):   # model.py:129-139, 86-99
        conv(ds + ".convLayer.0", (256, cin, 5, 5))
        norm(ds + ".convLayer.1", 256)
        conv(ds + ".convLayer_gates.0", (256, cin, 5, 5))
        norm(ds + ".convLayer_gates.1", 256)
    conv("conv2dto1dLayer", (256, 5120, 1))              # model.py:142-146
    norm("conv2dto1dLayer_tfan", 256)                    # model.py:147-148
    for i in range(1, 7):                                # model.py:151-180, 47-69
        r = "residualLayer%d" % i
        conv(r + ".conv1d_layer.0", (512, 256, 3))
        norm(r + ".conv1d_layer.1", 512)
        conv(r + ".conv_layer_gates.0", (512, 256, 3))
        norm(r + ".conv_layer_gates.1", 512)
        conv(r + ".conv1d_out_layer.0", (256, 512, 3))
        norm(r + ".conv1d_out_layer.1", 256)
    conv("conv1dto2dLayer", (5120, 256, 1))              # model.py:183-187
    norm("conv1dto2dLayer_tfan", 5120)                   # model.py:188-189
    conv("convLayer.0", (512, 256, 5, 5))                # alias of upSample2 (model.py:227)
    norm("convLayer.2", 128)
    conv("upSample1.0", (1024, 256, 5, 5))               # model.py:192-196
    norm("upSample1.2", 256)
    conv("upSample2.0", (512, 256, 5, 5))                # model.py:200-204
    norm("upSample2.2", 128)
    conv("lastConvLayer", (1, 128, 5, 15))               # model.py:207-211
    return ks


GEN_ALIASES = {  # state_dict key -> canonical (named_parameters) key it shares storage with
    "upSample2.0.weight": "convLayer.0.weight", "upSample2.0.bias": "convLayer.0.bias",
    "upSample2.2.weight": "convLayer.2.weight", "upSample2.2.bias": "convLayer.2.bias",
}


def discriminator_key_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    """The 20 state_dict entries of ``Discriminator()`` (model.py:290-327), dead ``downSample4`` included."""
    ks: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, shape):
        ks[name + ".weight"] = tuple(shape)
        ks[name + ".bias"] = (shape[0],)

    conv("convLayer1.0", (128, 1, 3, 3))
    for i, (co, ci) in enumerate(((256, 128), (512, 256), (1024, 512)), start=1):
        conv("downSample%d.0" % i, (co, ci, 3, 3))
        ks["downSample%d.1.weight" % i] = (co,)
        ks["downSample%d.1.bias" % i] = (co,)
    conv("downSample4.0", (1024, 1024, 1, 10))           # constructed, never used in forward (model.py:316-320, 340-349)
    ks["downSample4.1.weight"] = (1024,)
    ks["downSample4.1.bias"] = (1024,)
    for i in (1, 2, 3, 4):
        D_NORM_KEYS.update(("downSample%d.1.weight" % i, "downSample%d.1.bias" % i))
    conv("outputConvLayer.0", (1, 1024, 1, 3))
    return ks


DISC_DEAD_PREFIX = "downSample4."


def generator_param_names() -> List[str]:
    """The 110 unique parameters in ``named_parameters()`` order (aliases reported as ``convLayer.*``)."""
    return [k for k in generator_key_shapes() if k not in GEN_ALIASES]


def discriminator_param_names() -> List[str]:
    return list(discriminator_key_shapes())


# ----------------------------------------------------------------------------------------------
# deterministic, torch-version-independent weight filler
# ----------------------------------------------------------------------------------------------

def _is_norm_key(kind: str, key: str) -> bool:
    if kind == "G":
        if not NORM_KEYS:
            generator_key_shapes()
        return key in NORM_KEYS
    if not D_NORM_KEYS:
        discriminator_key_shapes()
    return key in D_NORM_KEYS


def filler_params(kind: str, seed: int, dtype=torch.float32) -> Params:
    """Fill every tensor from ``numpy.random.RandomState`` in state_dict order.

    Conv weight/bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (same law as torch's default
    ``reset_parameters``); InstanceNorm weight = 1 + U(-.25,.25), bias = U(-.25,.25) so the affine
    path is exercised.  Aliased generator keys share one tensor, like the reference.
    """
    shapes = generator_key_shapes() if kind == "G" else discriminator_key_shapes()
    rs = np.random.RandomState(seed)
    out: Params = OrderedDict()
    fan_in = None
    for key, shape in shapes.items():
        if key in GEN_ALIASES:
            out[key] = out[GEN_ALIASES[key]]
            continue
        if len(shape) > 1:                                 # conv weight
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            arr = rs.uniform(-bound, bound, size=shape)
        elif _is_norm_key(kind, key):
            arr = rs.uniform(-0.25, 0.25, size=shape)
            if key.endswith(".weight"):
                arr = arr + 1.0
        else:                                              # conv bias (follows its weight)
            bound = 1.0 / math.sqrt(fan_in)
            arr = rs.uniform(-bound, bound, size=shape)
        out[key] = torch.from_numpy(arr.astype(np.float32)).to(dtype)
    return out


def fif_mask(rs: np.random.RandomState, batch: int, n_mel: int, n_frames: int, max_mask_len: int) -> np.ndarray:
    """Filling-in-frames mask draw of dataset/vc_dataset.py:51-55: ones with [start,start+size) zeroed."""
    m = np.ones((batch, n_mel, n_frames), dtype=np.float32)
    for b in range(batch):
        size = rs.randint(0, max_mask_len)
        start = rs.randint(0, n_frames - size)
        m[b, :, start:start + size] = 0.0
    return m


# ----------------------------------------------------------------------------------------------
# model arithmetic
# ----------------------------------------------------------------------------------------------

def _inorm(x: torch.Tensor, p: Params, name: str) -> torch.Tensor:
    return F.instance_norm(x, None, None, p[name + ".weight"], p[name + ".bias"], True, 0.0, IN_EPS)


def silu_glu(x: torch.Tensor) -> torch.Tensor:
    """The reference's ``GLU`` is x*sigmoid(x) (model.py:20-21), not a channel-halving GLU."""
    return x * torch.sigmoid(x)


def generator_forward(p: Params, x: torch.Tensor, mask: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """``Generator.forward`` (model.py:239-280).  x, mask: [B,80,T] -> [B,80,T'] (T'=T when T%4==0)."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    h = torch.stack((x * mask, mask), dim=1)                                           # :241
    h = F.conv2d(h, p["conv1.weight"], p["conv1.bias"], 1, (2, 7)) * torch.sigmoid(
        F.conv2d(h, p["conv1_gates.weight"], p["conv1_gates.bias"], 1, (2, 7)))         # :242
    tap("conv1_glu", h)
    for ds in ("downSample1", "downSample2"):                                          # :245-246, :101-103
        a = _inorm(F.conv2d(h, p[ds + ".convLayer.0.weight"], p[ds + ".convLayer.0.bias"], 2, 2), p, ds + ".convLayer.1")
        g = _inorm(F.conv2d(h, p[ds + ".convLayer_gates.0.weight"], p[ds + ".convLayer_gates.0.bias"], 2, 2), p, ds + ".convLayer_gates.1")
        h = tap(ds, a * torch.sigmoid(g))
    B = h.shape[0]
    h = h.reshape(B, h.shape[1] * h.shape[2], -1)                                       # :249-251 (channel = c*H + h)
    h = _inorm(F.conv1d(h, p["conv2dto1dLayer.weight"], p["conv2dto1dLayer.bias"]), p, "conv2dto1dLayer_tfan")  # :254-255
    tap("conv2dto1d", h)
    for i in range(1, 7):                                                              # :258-263, :71-76
        r = "residualLayer%d" % i
        a = _inorm(F.conv1d(h, p[r + ".conv1d_layer.0.weight"], p[r + ".conv1d_layer.0.bias"], 1, 1), p, r + ".conv1d_layer.1")
        g = _inorm(F.conv1d(h, p[r + ".conv_layer_gates.0.weight"], p[r + ".conv_layer_gates.0.bias"], 1, 1), p, r + ".conv_layer_gates.1")
        o = _inorm(F.conv1d(a * torch.sigmoid(g), p[r + ".conv1d_out_layer.0.weight"], p[r + ".conv1d_out_layer.0.bias"], 1, 1), p, r + ".conv1d_out_layer.1")
        h = tap(r, h + o)
    h = _inorm(F.conv1d(h, p["conv1dto2dLayer.weight"], p["conv1dto2dLayer.bias"]), p, "conv1dto2dLayer_tfan")  # :266-267
    h = tap("conv1dto2d", h.reshape(B, 256, 20, -1))                                    # :270-271 (hard-coded 256, 20)
    for up in ("upSample1", "upSample2"):                                              # :274-275, :226-237
        h = F.pixel_shuffle(F.conv2d(h, p[up + ".0.weight"], p[up + ".0.bias"], 1, 2), 2)
        h = tap(up, silu_glu(_inorm(h, p, up + ".2")))
    h = F.conv2d(h, p["lastConvLayer.weight"], p["lastConvLayer.bias"], 1, (2, 7))      # :278
    return h.squeeze(1)                                                                 # :279


def discriminator_forward(p: Params, x: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """``Discriminator.forward`` (model.py:340-349).  x: [B,80,T] -> [B,1,10,T/8] in (0,1)."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    h = x.unsqueeze(1)                                                                  # :343
    h = tap("convLayer1", silu_glu(F.conv2d(h, p["convLayer1.0.weight"], p["convLayer1.0.bias"], 1, 1)))  # :344
    for i in (1, 2, 3):                                                                # :345-347, :329-338
        d = "downSample%d" % i
        h = F.conv2d(h, p[d + ".0.weight"], p[d + ".0.bias"], 2, 1)
        h = tap(d, silu_glu(_inorm(h, p, d + ".1")))
    h = F.conv2d(h, p["outputConvLayer.0.weight"], p["outputConvLayer.0.bias"], 1, (0, 1))
    return torch.sigmoid(h)                                                             # :348


# ----------------------------------------------------------------------------------------------
# training step (train.py:195-315)
# ----------------------------------------------------------------------------------------------

NET_ORDER = ("generator_A2B", "generator_B2A", "discriminator_A", "discriminator_B",
             "discriminator_A2", "discriminator_B2")        # construction order, train.py:103-110


class AdamState:
    """torch.optim.Adam(betas=(0.5,0.999), eps=1e-8, weight_decay=0) restated (train.py:119-122).

    Parameters whose grad is None (D ``downSample4``) are skipped and acquire no state, exactly
    like torch.optim.Adam.
    """

    def __init__(self, params: List[torch.Tensor], lr: float, betas=(0.5, 0.999), eps=1e-8):
        self.params = params
        self.lr = lr
        self.b1, self.b2 = betas
        self.eps = eps
        self.state: Dict[int, dict] = {}

    def step(self, grads: List[torch.Tensor | None]):
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if g is None:
                continue
            st = self.state.setdefault(i, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            st["m"].lerp_(g, 1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-self.lr / bc1)


class StepOracle:
    """One MaskCycleGAN-VC G+D iteration on CPU, semantics of train.py:195-315.

    ``nets`` maps the six net names to Params dicts.  The G optimizer owns the 2x110 unique
    generator tensors, the D optimizer the 4x20 discriminator tensors (train.py:113-122).
    ``skip_wasted=True`` drops work whose results the reference discards (D-parameter grads in the
    G step, G backward in the D step) -- mathematically invisible; used only to time a fairer
    cpu_baseline and to prove the invisibility in tests.
    """

    def __init__(self, nets: Dict[str, Params], g_lr=2e-4, d_lr=1e-4, cycle_lambda=10.0, identity_lambda=5.0,
                 skip_wasted: bool = False):
        self.nets = nets
        self.cycle_lambda = cycle_lambda
        self.identity_lambda = identity_lambda
        self.skip_wasted = skip_wasted
        gnames, dnames = generator_param_names(), discriminator_param_names()
        self.g_list = [nets[n][k] for n in NET_ORDER[:2] for k in gnames]
        self.d_list = [nets[n][k] for n in NET_ORDER[2:] for k in dnames]
        self.g_opt = AdamState(self.g_list, g_lr)
        self.d_opt = AdamState(self.d_list, d_lr)

    def _G(self, name, x, m):
        return generator_forward(self.nets[name], x, m)

    def _D(self, name, x):
        return discriminator_forward(self.nets[name], x)

    def losses_g(self, real_A, mask_A, real_B, mask_B):
        """Generator-phase forward + loss (train.py:203-237). Returns (g_loss, dict of terms/tensors)."""
        fake_B = self._G("generator_A2B", real_A, mask_A)
        cycle_A = self._G("generator_B2A", fake_B, torch.ones_like(fake_B))
        fake_A = self._G("generator_B2A", real_B, mask_B)
        cycle_B = self._G("generator_A2B", fake_A, torch.ones_like(fake_A))
        identity_A = self._G("generator_B2A", real_A, torch.ones_like(real_A))
        identity_B = self._G("generator_A2B", real_B, torch.ones_like(real_B))
        d_fake_A = self._D("discriminator_A", fake_A)
        d_fake_B = self._D("discriminator_B", fake_B)
        d_fake_cycle_A = self._D("discriminator_A2", cycle_A)
        d_fake_cycle_B = self._D("discriminator_B2", cycle_B)
        cycle = torch.mean(torch.abs(real_A - cycle_A)) + torch.mean(torch.abs(real_B - cycle_B))
        ident = torch.mean(torch.abs(real_A - identity_A)) + torch.mean(torch.abs(real_B - identity_B))
        adv = (torch.mean((1 - d_fake_B) ** 2) + torch.mean((1 - d_fake_A) ** 2)
               + torch.mean((1 - d_fake_cycle_B) ** 2) + torch.mean((1 - d_fake_cycle_A) ** 2))
        g_loss = adv + self.cycle_lambda * cycle + self.identity_lambda * ident
        return g_loss, dict(fake_A=fake_A, fake_B=fake_B, cycle_A=cycle_A, cycle_B=cycle_B,
                            identity_A=identity_A, identity_B=identity_B, cycle_loss=cycle,
                            identity_loss=ident, adv_loss=adv)

    def losses_d(self, real_A, mask_A, real_B, mask_B):
        """Discriminator-phase forward + loss (train.py:255-294); generators run with post-update weights."""
        d_real_A = self._D("discriminator_A", real_A)
        d_real_B = self._D("discriminator_B", real_B)
        d_real_A2 = self._D("discriminator_A2", real_A)
        d_real_B2 = self._D("discriminator_B2", real_B)
        generated_A = self._G("generator_B2A", real_B, mask_B)
        d_fake_A = self._D("discriminator_A", generated_A)
        cycled_B = self._G("generator_A2B", generated_A, torch.ones_like(generated_A))
        d_cycled_B = self._D("discriminator_B2", cycled_B)
        generated_B = self._G("generator_A2B", real_A, mask_A)
        d_fake_B = self._D("discriminator_B", generated_B)
        cycled_A = self._G("generator_B2A", generated_B, torch.ones_like(generated_B))
        d_cycled_A = self._D("discriminator_A2", cycled_A)
        d_loss_A = (torch.mean((1 - d_real_A) ** 2) + torch.mean(d_fake_A ** 2)) / 2.0
        d_loss_B = (torch.mean((1 - d_real_B) ** 2) + torch.mean(d_fake_B ** 2)) / 2.0
        d_loss_A_2nd = (torch.mean((1 - d_real_A2) ** 2) + torch.mean(d_cycled_A ** 2)) / 2.0
        d_loss_B_2nd = (torch.mean((1 - d_real_B2) ** 2) + torch.mean(d_cycled_B ** 2)) / 2.0
        return (d_loss_A + d_loss_B) / 2.0 + (d_loss_A_2nd + d_loss_B_2nd) / 2.0

    def step(self, real_A, mask_A, real_B, mask_B, return_grads: bool = False):
        """Returns (g_loss, d_loss) floats [+ (g_grads, d_grads)].  Updates parameters in place."""
        # ---- generator phase (train.py:195-242)
        for t in self.g_list:
            t.requires_grad_(True)
        for t in self.d_list:
            t.requires_grad_(not self.skip_wasted)
        g_loss, _ = self.losses_g(real_A, mask_A, real_B, mask_B)
        g_grads = list(torch.autograd.grad(g_loss, self.g_list, allow_unused=True))
        for t in self.g_list + self.d_list:
            t.requires_grad_(False)
        with torch.no_grad():
            self.g_opt.step(g_grads)
        # ---- discriminator phase (train.py:247-299)
        for t in self.d_list:
            t.requires_grad_(True)
        if self.skip_wasted:
            d_loss = self.losses_d(real_A, mask_A, real_B, mask_B)
        else:
            for t in self.g_list:
                t.requires_grad_(True)           # the reference does not detach the generated tensors
            d_loss = self.losses_d(real_A, mask_A, real_B, mask_B)
        d_grads = list(torch.autograd.grad(d_loss, self.d_list, allow_unused=True))
        for t in self.g_list + self.d_list:
            t.requires_grad_(False)
        with torch.no_grad():
            self.d_opt.step(d_grads)
        if return_grads:
            return float(g_loss), float(d_loss), g_grads, d_grads
        return float(g_loss), float(d_loss)


def sample_index(numel: int, k: int = 256) -> np.ndarray:
    """Positions of the parameter samples committed with the multi-step train() fixtures (tests/golden/make_golden.py): up to ``k``
    seeded positions per tensor, every element of a smaller tensor."""
    if numel <= k:
        return np.arange(numel)
    return np.unique(np.random.RandomState(numel % 100003).randint(0, numel, k))


def default_init_nets(seed: int = 0) -> Dict[str, Params]:
    """Six nets with torch's default init law drawn from ``torch.manual_seed(seed)`` in the reference's
    construction order (train.py:103-110).  Uses nn.Conv*/InstanceNorm reset laws via plain tensors:
    kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight, same bound for bias.
    NOTE: RNG stream consumption matches torch's ``reset_parameters`` order (weight then bias per conv).
    """
    torch.manual_seed(seed)
    nets: Dict[str, Params] = OrderedDict()
    for name in NET_ORDER:
        shapes = generator_key_shapes() if name.startswith("generator") else discriminator_key_shapes()
        # the reference constructs upSample1 before upSample2 (model.py:192-204); the alias key
        # ``convLayer.*`` appears first in state_dict order but is initialised last of the two.
        order = list(shapes)
        if name.startswith("generator"):
            order = ([k for k in order if not k.startswith(("convLayer.", "upSample", "lastConvLayer"))]
                     + [k for k in order if k.startswith("upSample1")]
                     + [k for k in order if k.startswith("convLayer.")]
                     + [k for k in order if k.startswith("lastConvLayer")])
        p: Params = {}
        fan_in = 1
        for key in order:
            shape = shapes[key]
            if len(shape) > 1:
                fan_in = int(np.prod(shape[1:]))
                # torch.nn.init.kaiming_uniform_(w, a=sqrt(5)): gain*sqrt(3/fan_in), computed the same way
                gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
                b = math.sqrt(3.0) * (gain / math.sqrt(fan_in))
                p[key] = torch.empty(shape).uniform_(-b, b)
            elif _is_norm_key("G" if name.startswith("generator") else "D", key):
                p[key] = torch.ones(shape) if key.endswith(".weight") else torch.zeros(shape)
            else:
                b = 1.0 / math.sqrt(fan_in)
                p[key] = torch.empty(shape).uniform_(-b, b)
        out: Params = OrderedDict()
        for key in shapes:
            out[key] = p[GEN_ALIASES.get(key, key)] if name.startswith("generator") else p[key]
        nets[name] = out
    return nets
