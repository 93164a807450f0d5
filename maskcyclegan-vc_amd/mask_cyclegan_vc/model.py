"""MaskCycleGAN-VC networks on MI355X (gfx950) HIP kernels -- drop-in for the reference's
``mask_cyclegan_vc/model.py``.

Same import path, class names, constructor signatures, ``nn.Module`` behaviour (``parameters()``,
``.to()``, ``state_dict()`` with the reference's 114 / 20 keys incl. the aliased ``convLayer.*`` and
the dead ``downSample4.*``) and autograd semantics; the arithmetic runs in ``libmcvc_hip.so``.

* ``Generator.forward`` / ``Discriminator.forward`` are ONE library call per pass (forward and
  backward each): the C++ side owns the layer schedule (csrc/net.hip).
* The building blocks (``ResidualLayer``, ``DownSampleGenerator``, ``GLU`` ...) run on the same
  kernels through the single-op entry points (``ops.py``).
* ``torch.nn`` layer objects are used purely as *parameter containers* so that key names, default
  initialisation law and RNG consumption order match the reference construction
  (reference model.py:110-211, 287-327); their ``forward`` is never called.

There is no CPU path: calling a module on CPU tensors raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _hip, ops
from ._hip import check, lib, ptr, ptr_table, stream

__all__ = ["GLU", "PixelShuffle", "ResidualLayer", "DownSampleGenerator", "Generator", "Discriminator"]


# ------------------------------------------------------------------------------------------------
# small blocks
# ------------------------------------------------------------------------------------------------
class GLU(nn.Module):
    """The reference's "GLU": ``x * sigmoid(x)`` with no channel halving (reference model.py:12-21)."""

    def forward(self, x):
        return ops.activation(x, ops.ACT_SILU)


class PixelShuffle(nn.Module):
    """3-D "pixel shuffle" kept for API parity (reference model.py:24-37, unused there as well):
    a pure view ``[N, C, W] -> [N, C/2, 2W]``."""

    def __init__(self, upscale_factor):
        super().__init__()
        self.upscale_factor = upscale_factor

    def forward(self, x):
        return x.view(x.shape[0], x.shape[1] // 2, x.shape[2] * 2)


def _conv_norm_1d(cin, cout, k, pad):
    return nn.Sequential(nn.Conv1d(cin, cout, k, stride=1, padding=pad), nn.InstanceNorm1d(cout, affine=True))


def _conv_norm_2d(cin, cout, k, stride, pad):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=pad), nn.InstanceNorm2d(cout, affine=True))


class ResidualLayer(nn.Module):
    """``x + IN(conv(IN(conv_a(x)) * sigmoid(IN(conv_g(x)))))`` (reference model.py:40-76).
    ``stride`` is accepted and ignored, like the reference (it hard-codes 1)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.conv1d_layer = _conv_norm_1d(in_channels, out_channels, kernel_size, padding)
        self.conv_layer_gates = _conv_norm_1d(in_channels, out_channels, kernel_size, padding)
        self.conv1d_out_layer = _conv_norm_1d(out_channels, in_channels, kernel_size, padding)
        self._pad = padding

    def forward(self, x):
        a, g, o = self.conv1d_layer, self.conv_layer_gates, self.conv1d_out_layer
        # value and gate convolutions share the input: one conv with concatenated output channels
        pre = ops.conv1d(x, torch.cat((a[0].weight, g[0].weight), 0), torch.cat((a[0].bias, g[0].bias), 0), self._pad)
        glu = ops.instnorm_act(pre.unsqueeze(2), a[1].weight, a[1].bias, ops.ACT_GLU, g[1].weight, g[1].bias).squeeze(2)
        out = ops.conv1d(glu, o[0].weight, o[0].bias, self._pad)
        return ops.instnorm_act(out.unsqueeze(2), o[1].weight, o[1].bias, ops.ACT_NONE, residual=x.contiguous().unsqueeze(2)).squeeze(2)


class DownSampleGenerator(nn.Module):
    """``IN(conv_a(x)) * sigmoid(IN(conv_g(x)))`` with a strided 2-D conv (reference model.py:79-103)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.convLayer = _conv_norm_2d(in_channels, out_channels, kernel_size, stride, padding)
        self.convLayer_gates = _conv_norm_2d(in_channels, out_channels, kernel_size, stride, padding)

    def forward(self, x):
        a, g = self.convLayer, self.convLayer_gates
        conv = a[0]
        pre = ops.conv2d(x, torch.cat((a[0].weight, g[0].weight), 0), torch.cat((a[0].bias, g[0].bias), 0),
                         conv.stride[0], conv.padding)
        return ops.instnorm_act(pre, a[1].weight, a[1].bias, ops.ACT_GLU, g[1].weight, g[1].bias)


# ------------------------------------------------------------------------------------------------
# whole-network calls
# ------------------------------------------------------------------------------------------------
class _NetBase(nn.Module):
    """Shared plumbing: pointer tables, packed-weight cache, workspace cache."""

    _kind = "gen"
    _nparams = 0

    def _init_runtime(self):
        self._packed = None
        self._packed_version = None
        self._ws = {}
        self._iws = {}
        self._bf16 = None
        self._bf16_version = None

    def _plist(self):
        ps = list(self.parameters())
        assert len(ps) == self._nparams, (len(ps), self._nparams)
        return ps

    def _param_version(self, ps):
        return tuple((p.data_ptr(), p._version) for p in ps)

    def packed_weights(self, ps=None, force=False):
        """K-major packed copy of the conv weights; refreshed whenever a parameter changed."""
        ps = ps or self._plist()
        _hip.require_cuda_f32(*ps)
        ver = self._param_version(ps)
        L = lib()
        if self._packed is None or self._packed.device != ps[0].device:
            n = L.mcvc_gen_packed_floats() if self._kind == "gen" else L.mcvc_disc_packed_floats()
            self._packed = torch.zeros(n, device=ps[0].device)
            self._packed_version = None
        if force or ver != self._packed_version:
            fn = L.mcvc_gen_pack if self._kind == "gen" else L.mcvc_disc_pack
            check(fn(ptr_table(ps), ptr(self._packed), stream()), "mcvc_%s_pack" % self._kind)
            self._packed_version = ver
        return self._packed

    def workspace(self, B, T, device):
        key = (B, T, str(device))
        ws = self._ws.get(key)
        if ws is None:
            L = lib()
            if self._kind == "gen":
                n_stash, n_scr = L.mcvc_gen_stash_floats(B, T), L.mcvc_gen_scratch_floats(B, T)
            else:
                n_stash, n_scr = L.mcvc_disc_stash_floats(B, T), L.mcvc_disc_scratch_floats(B, T)
            ws = (n_stash, torch.empty(n_scr, device=device))
            self._ws = {key: ws}            # keep one shape resident (drops the previous shape's scratch and inference stash)
        return ws


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, mask, *params):
        x = x.contiguous()
        mask = mask.contiguous()
        _hip.require_cuda_f32(x, mask)
        B, n_mel, T = x.shape
        if n_mel != _hip.N_MEL:
            raise RuntimeError("Generator expects %d mel bins (reference model.py:271 hard-codes 256x20)" % _hip.N_MEL)
        L = lib()
        packed = net.packed_weights(list(params))
        n_stash, scratch = net.workspace(B, T, x.device)
        stash = torch.empty(n_stash, device=x.device)
        out = torch.empty((B, n_mel, L.mcvc_gen_out_frames(T)), device=x.device)
        check(L.mcvc_gen_forward(ptr_table(params), ptr(packed), ptr(x), ptr(mask), ptr(out), ptr(stash), ptr(scratch), scratch.numel(),
                                 B, T, stream()), "mcvc_gen_forward")
        ctx.net = net
        ctx.dims = (B, T)
        ctx.save_for_backward(mask, stash, packed, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        mask, stash, packed, *params = ctx.saved_tensors
        B, T = ctx.dims
        dout = dout.contiguous()
        L = lib()
        _, scratch = ctx.net.workspace(B, T, dout.device)
        need = ctx.needs_input_grad
        sizes = [p.numel() if need[3 + i] else 0 for i, p in enumerate(params)]
        flat = torch.zeros(sum((n + 3) & ~3 for n in sizes), device=dout.device)
        grads, off = [], 0
        for p, n in zip(params, sizes):
            grads.append(flat[off:off + n].view_as(p) if n else None)
            off += (n + 3) & ~3          # 16-byte aligned views
        dx = torch.empty((B, _hip.N_MEL, T), device=dout.device) if need[1] else None
        check(L.mcvc_gen_backward(ptr_table(params), ptr(packed), ptr_table(grads) if any(sizes) else None, ptr(mask), ptr(dout), ptr(dx), 0,
                                  ptr(stash), ptr(scratch), scratch.numel(), B, T, stream(), None), "mcvc_gen_backward")
        return (None, dx, None, *grads)


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        x = x.contiguous()
        _hip.require_cuda_f32(x)
        B, n_mel, T = x.shape
        if n_mel != _hip.N_MEL:
            raise RuntimeError("Discriminator expects %d mel bins" % _hip.N_MEL)
        L = lib()
        packed = net.packed_weights(list(params))
        n_stash, scratch = net.workspace(B, T, x.device)
        stash = torch.empty(n_stash, device=x.device)
        out = torch.empty((B, 1, 10, L.mcvc_disc_out_frames(T)), device=x.device)
        check(L.mcvc_disc_forward(ptr_table(params), ptr(packed), ptr(x), ptr(out), ptr(stash), ptr(scratch), scratch.numel(), B, T, stream()),
              "mcvc_disc_forward")
        ctx.net = net
        ctx.dims = (B, T)
        ctx.save_for_backward(stash, packed, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        stash, packed, *params = ctx.saved_tensors
        B, T = ctx.dims
        dout = dout.contiguous()
        L = lib()
        _, scratch = ctx.net.workspace(B, T, dout.device)
        need = ctx.needs_input_grad
        # downSample4 (parameter slots 14..17) takes no part in forward: its grads stay None, like the reference
        sizes = [p.numel() if (need[2 + i] and not 14 <= i <= 17) else 0 for i, p in enumerate(params)]
        flat = torch.zeros(sum((n + 3) & ~3 for n in sizes), device=dout.device)
        grads, off = [], 0
        for p, n in zip(params, sizes):
            grads.append(flat[off:off + n].view_as(p) if n else None)
            off += (n + 3) & ~3
        dx = torch.empty((B, _hip.N_MEL, T), device=dout.device) if need[1] else None
        check(L.mcvc_disc_backward(ptr_table(params), ptr(packed), ptr_table(grads) if any(sizes) else None, ptr(dout), 0, ptr(dx), 0,
                                   ptr(stash), ptr(scratch), scratch.numel(), B, T, stream(), None), "mcvc_disc_backward")
        return (None, dx, *grads)


class Generator(_NetBase):
    """MaskCycleGAN-VC generator (reference model.py:106-280).

    ``forward(x, mask)``: ``x, mask`` float32 ``[B, 80, T]`` -> ``[B, 80, T]`` (T a multiple of 4
    keeps the length, like the reference)."""

    _kind = "gen"
    _nparams = _hip.GEN_NPARAMS

    def __init__(self, input_shape=(80, 64), residual_in_channels=256):
        super().__init__()
        n_mel, _ = input_shape
        ch = residual_in_channels
        if n_mel != 80 or ch != 256:
            # the reference forward hard-codes view(B, 256, 20, -1) (model.py:271): only these values work there too
            raise ValueError("Generator supports input_shape=(80, T) and residual_in_channels=256")
        self.flattened_channels = (n_mel // 4) * ch
        # --- construction order == reference order, so seeded default init matches bit for bit
        self.conv1 = nn.Conv2d(2, ch // 2, (5, 15), stride=(1, 1), padding=(2, 7))
        self.conv1_gates = nn.Conv2d(2, ch // 2, (5, 15), stride=1, padding=(2, 7))
        self.downSample1 = DownSampleGenerator(ch // 2, ch, 5, 2, 2)
        self.downSample2 = DownSampleGenerator(ch, ch, 5, 2, 2)
        self.conv2dto1dLayer = nn.Conv1d(self.flattened_channels, ch, 1, stride=1, padding=0)
        self.conv2dto1dLayer_tfan = nn.InstanceNorm1d(ch, affine=True)
        for i in range(1, 7):
            setattr(self, "residualLayer%d" % i, ResidualLayer(ch, ch * 2, 3, 1, 1))
        self.conv1dto2dLayer = nn.Conv1d(ch, self.flattened_channels, 1, stride=1, padding=0)
        self.conv1dto2dLayer_tfan = nn.InstanceNorm1d(self.flattened_channels, affine=True)
        self.upSample1 = self.upsample(ch, ch * 4, 5, 1, 2)
        self.glu = GLU()
        self.upSample2 = self.upsample(ch, ch * 2, 5, 1, 2)
        self.lastConvLayer = nn.Conv2d(ch // 2, 1, (5, 15), stride=(1, 1), padding=(2, 7))
        self._init_runtime()

    def upsample(self, in_channels, out_channels, kernel_size, stride, padding):
        """conv -> PixelShuffle(2) -> InstanceNorm -> x*sigmoid(x).  Like the reference
        (model.py:226-237) the result is also bound to ``self.convLayer``, which is what makes
        ``convLayer.*`` alias ``upSample2.*`` in ``state_dict()``."""
        self.convLayer = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding),
                                       nn.PixelShuffle(upscale_factor=2),
                                       nn.InstanceNorm2d(out_channels // 4, affine=True),
                                       GLU())
        return self.convLayer

    def downsample(self, in_channels, out_channels, kernel_size, stride, padding):
        """Unused factory kept for API parity (reference model.py:213-224)."""
        self.ConvLayer = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding),
                                       nn.InstanceNorm1d(out_channels, affine=True), GLU())
        return self.ConvLayer

    def forward(self, x, mask):
        return _GeneratorFn.apply(self, x, mask, *self._plist())

    def _infer_workspace(self, B, T, device):
        """(stash, scratch) of the gradient-free forward, one set per (shape, HIP stream): the bucketed inference driver runs
        forwards of different shapes on two streams at once, which must not share transient buffers."""
        key = (B, T, str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._iws.get(key)
        if ws is None:
            L = lib()
            while len(self._iws) >= 4:
                self._iws.pop(next(iter(self._iws)))
            ws = (torch.empty(L.mcvc_gen_stash_floats(B, T), device=device), torch.empty(L.mcvc_gen_scratch_floats(B, T), device=device))
            self._iws[key] = ws
        return ws

    def prepare_inference(self, dtype="f32"):
        """Build / refresh the packed weights ``infer`` reads, on the current stream (call before forking inference streams)."""
        if dtype == "f32":
            self.packed_weights()
        elif dtype == "bf16":
            self._bf16_pack()
        else:
            raise ValueError("dtype must be 'f32' or 'bf16'")

    def _bf16_pack(self, ps=None):
        """bf16 weight pack of the inference forward (cast from the fp32 parameters); refreshed when a parameter changed."""
        ps = ps or self._plist()
        _hip.require_cuda_f32(*ps)
        ver = self._param_version(ps)
        L = lib()
        if self._bf16 is None or self._bf16.device != ps[0].device:
            self._bf16 = torch.zeros(L.mcvc_gen_bf16_packed_bytes(), dtype=torch.uint8, device=ps[0].device)
            self._bf16_version = None
        if ver != self._bf16_version:
            check(L.mcvc_gen_bf16_pack(ptr_table(ps), ptr(self._bf16), stream()), "mcvc_gen_bf16_pack")
            self._bf16_version = ver
        return self._bf16

    def infer(self, x, mask=None, dtype="f32"):
        """Gradient-free forward for the inference driver (reference test.py:85-119 calls ``generator(real, ones_like(real))``
        and never back-propagates): no autograd node, no fresh stash allocation per call.  ``mask=None`` is the all-ones
        mask of test.py:92.  ``dtype``: "f32" (bit-for-bit the training forward) or "bf16" (BASELINE configs[4]: bf16 MFMA
        with fp32 accumulation and fp32 InstanceNorm statistics; weights are cast from the fp32 parameters)."""
        x = x.contiguous()
        _hip.require_cuda_f32(x, mask)
        B, n_mel, T = x.shape
        if n_mel != _hip.N_MEL:
            raise RuntimeError("Generator expects %d mel bins" % _hip.N_MEL)
        L = lib()
        ps = self._plist()
        out = torch.empty((B, n_mel, L.mcvc_gen_out_frames(T)), device=x.device)
        if dtype == "f32":
            packed = self.packed_weights(ps)
            stash, scratch = self._infer_workspace(B, T, x.device)
            check(L.mcvc_gen_forward(ptr_table(ps), ptr(packed), ptr(x), ptr(mask.contiguous() if mask is not None else None), ptr(out),
                                     ptr(stash), ptr(scratch), scratch.numel(), B, T, stream()), "mcvc_gen_forward")
            return out
        if dtype == "bf16":
            packed = self._bf16_pack(ps)
            key = ("bf16", B, T, str(x.device), torch.cuda.current_stream(x.device).cuda_stream)
            ws = self._iws.get(key)
            if ws is None:
                while len(self._iws) >= 4:
                    self._iws.pop(next(iter(self._iws)))
                ws = torch.empty(L.mcvc_gen_bf16_workspace_bytes(B, T), dtype=torch.uint8, device=x.device)
                self._iws[key] = ws
            check(L.mcvc_gen_infer_bf16(ptr_table(ps), ptr(packed), ptr(x), ptr(mask.contiguous() if mask is not None else None), ptr(out),
                                        ptr(ws), ws.numel(), B, T, stream()), "mcvc_gen_infer_bf16")
            return out
        raise ValueError("dtype must be 'f32' or 'bf16'")


class Discriminator(_NetBase):
    """PatchGAN discriminator (reference model.py:283-349): ``[B, 80, T] -> [B, 1, 10, T/8]`` in (0, 1)."""

    _kind = "disc"
    _nparams = _hip.DISC_NPARAMS

    def __init__(self, input_shape=(80, 64), residual_in_channels=256):
        super().__init__()
        ch = residual_in_channels
        if ch != 256:
            raise ValueError("Discriminator supports residual_in_channels=256")
        self.convLayer1 = nn.Sequential(nn.Conv2d(1, ch // 2, (3, 3), stride=(1, 1), padding=(1, 1)), GLU())
        self.downSample1 = self.downsample(ch // 2, ch, (3, 3), (2, 2), 1)
        self.downSample2 = self.downsample(ch, ch * 2, (3, 3), (2, 2), 1)
        self.downSample3 = self.downsample(ch * 2, ch * 4, (3, 3), (2, 2), 1)
        # constructed, check-pointed and handed to Adam by the reference, but never used in forward
        # (model.py:316-320 vs :340-349): kept so parameter counts / state_dict / optimizer indices match
        self.downSample4 = self.downsample(ch * 4, ch * 4, (1, 10), (1, 1), (0, 2))
        self.outputConvLayer = nn.Sequential(nn.Conv2d(ch * 4, 1, (1, 3), stride=(1, 1), padding=(0, 1)))
        self._init_runtime()

    def downsample(self, in_channels, out_channels, kernel_size, stride, padding):
        return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding),
                             nn.InstanceNorm2d(out_channels, affine=True), GLU())

    def forward(self, x):
        return _DiscriminatorFn.apply(self, x, *self._plist())
