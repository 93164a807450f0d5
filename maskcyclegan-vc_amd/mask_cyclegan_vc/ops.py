"""Autograd-visible single operators on the gfx950 kernels (thin wrappers over include/mcvc.h).

These back the stand-alone building blocks of ``model.py`` (``ResidualLayer``,
``DownSampleGenerator``, ``GLU`` ...) and the kernel parity tests.  ``Generator`` / ``Discriminator``
do not go through here: they call the whole-network entry points (one host call per pass).
"""
from __future__ import annotations

import torch

from . import _hip
from ._hip import check, lib, ptr, stream

ACT_NONE, ACT_GLU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3
MAX_SLABS = 16


def _out_hw(H, W, KH, KW, stride, ph, pw):
    return (H + 2 * ph - KH) // stride + 1, (W + 2 * pw - KW) // stride + 1


def conv2d_forward(x, w, b, stride, padding, pixel_shuffle=False):
    _hip.require_cuda_f32(x, w, b)
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    ph, pw = padding
    OH, OW = _out_hw(H, W, KH, KW, stride, ph, pw)
    L = lib()
    wpack = torch.zeros(L.mcvc_conv2d_pack_floats(Cout, Cin, KH, KW), device=x.device)
    y = torch.empty((N, Cout // 4, 2 * OH, 2 * OW) if pixel_shuffle else (N, Cout, OH, OW), device=x.device)
    slabs = torch.empty((MAX_SLABS - 1) * y.numel(), device=x.device)
    check(L.mcvc_conv2d_forward(ptr(x), ptr(w), ptr(b), ptr(y), ptr(wpack), ptr(slabs), MAX_SLABS, N, Cin, H, W, Cout, KH, KW,
                                stride, ph, pw, int(pixel_shuffle), stream()), "mcvc_conv2d_forward")
    return y


def conv2d_dgrad(dy, w, x_shape, stride, padding):
    _hip.require_cuda_f32(dy, w)
    N, Cin, H, W = x_shape
    Cout, _, KH, KW = w.shape
    L = lib()
    wpack = torch.zeros(L.mcvc_conv2d_pack_floats(Cout, Cin, KH, KW), device=dy.device)
    dx = torch.empty(x_shape, device=dy.device)
    slabs = torch.empty((MAX_SLABS - 1) * dx.numel(), device=dy.device)
    check(L.mcvc_conv2d_dgrad(ptr(dy), ptr(w), ptr(dx), ptr(wpack), ptr(slabs), MAX_SLABS, N, Cin, H, W, Cout, KH, KW,
                              stride, padding[0], padding[1], stream()), "mcvc_conv2d_dgrad")
    return dx


def conv2d_wgrad(x, dy, w_shape, stride, padding):
    _hip.require_cuda_f32(x, dy)
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w_shape
    dw = torch.zeros(w_shape, device=x.device)
    L = lib()
    n_slab = L.mcvc_conv2d_wgrad_slab_floats(N, Cin, H, W, Cout, KH, KW, stride, padding[0], padding[1])
    slabs = torch.empty(n_slab, device=x.device) if n_slab > 0 else None
    check(L.mcvc_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(slabs), n_slab, N, Cin, H, W, Cout, KH, KW, stride, padding[0], padding[1], stream()),
          "mcvc_conv2d_wgrad")
    return dw


def bias_grad(dy):
    N, C = dy.shape[:2]
    P = dy[0, 0].numel()
    db = torch.zeros(C, device=dy.device)
    check(lib().mcvc_bias_grad(ptr(dy), ptr(db), N, C, P, stream()), "mcvc_bias_grad")
    return db


def _unshuffle(dy):
    """[N, C, 2H, 2W] -> [N, 4C, H, W] (inverse of PixelShuffle(2)); a strided copy, plumbing only."""
    N, C, H2, W2 = dy.shape
    return dy.view(N, C, H2 // 2, 2, W2 // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(N, 4 * C, H2 // 2, W2 // 2).contiguous()


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, pixel_shuffle):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, pixel_shuffle)
        return conv2d_forward(x, w, b, stride, padding, pixel_shuffle)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, pixel_shuffle = ctx.cfg
        dy = dy.contiguous()
        if pixel_shuffle:
            dy = _unshuffle(dy)
        dx = conv2d_dgrad(dy, w, tuple(x.shape), stride, padding) if ctx.needs_input_grad[0] else None
        dw = conv2d_wgrad(x, dy, tuple(w.shape), stride, padding) if ctx.needs_input_grad[1] else None
        db = bias_grad(dy) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None


def conv2d(x, w, b, stride=1, padding=(0, 0), pixel_shuffle=False):
    return _Conv2d.apply(x, w, b, stride, tuple(padding), pixel_shuffle)


def conv1d(x, w, b, padding=0):
    """[N,C,T] conv as a 1xK 2-D conv (the kernels treat Conv1d as KH=1)."""
    y = conv2d(x.unsqueeze(2), w.unsqueeze(2), b, 1, (0, padding))
    return y.squeeze(2)


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, gamma_g, beta_g, residual, act):
        x = x.contiguous()
        _hip.require_cuda_f32(x, gamma, beta, gamma_g, beta_g, residual)
        N, Cx, H, W = x.shape
        C = Cx // 2 if act == ACT_GLU else Cx
        y = torch.empty((N, C, H, W), device=x.device)
        stats = torch.empty((N, Cx, 2), device=x.device)
        check(lib().mcvc_instnorm_act_forward(ptr(x), ptr(gamma), ptr(beta), ptr(gamma_g), ptr(beta_g), ptr(residual), ptr(y), ptr(stats),
                                              N, C, H, W, act, stream()), "mcvc_instnorm_act_forward")
        ctx.save_for_backward(x, gamma, beta, gamma_g, beta_g, stats)
        ctx.act = act
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, gamma_g, beta_g, stats = ctx.saved_tensors
        act = ctx.act
        N, Cx, H, W = x.shape
        C = Cx // 2 if act == ACT_GLU else Cx
        dy = dy.contiguous().clone()            # the kernel may reduce slabs into dy in place
        dx = torch.empty_like(x)
        dg = torch.zeros_like(gamma); db = torch.zeros_like(beta)
        dgg = torch.zeros_like(gamma_g) if gamma_g is not None else None
        dbg = torch.zeros_like(beta_g) if beta_g is not None else None
        check(lib().mcvc_instnorm_act_backward(ptr(x), ptr(gamma), ptr(beta), ptr(gamma_g), ptr(beta_g), ptr(stats), ptr(dy), ptr(dx),
                                               ptr(dg), ptr(db), ptr(dgg), ptr(dbg), N, C, H, W, act, stream()), "mcvc_instnorm_act_backward")
        return dx, dg, db, dgg, dbg, (dy if ctx.has_res else None), None


def instnorm_act(x, gamma, beta, act=ACT_NONE, gamma_gate=None, beta_gate=None, residual=None):
    """InstanceNorm(affine) over [N,Cx,H,W] (+ activation, + residual)."""
    return _InstNormAct.apply(x, gamma, beta, gamma_gate, beta_gate, residual, act)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous().clone()
        _hip.require_cuda_f32(x)
        N, Cx = x.shape[:2]
        P = x[0, 0].numel()
        C = Cx // 2 if act == ACT_GLU else Cx
        y = torch.empty((N, C) + tuple(x.shape[2:]), device=x.device)
        check(lib().mcvc_act_forward(ptr(x), ptr(y), N, C, P, act, stream()), "mcvc_act_forward")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        act = ctx.act
        N, Cx = x.shape[:2]
        P = x[0, 0].numel()
        C = Cx // 2 if act == ACT_GLU else Cx
        dy = dy.contiguous().clone()
        dx = torch.empty_like(x)
        check(lib().mcvc_act_backward(ptr(x), ptr(dy), ptr(dx), N, C, P, act, stream()), "mcvc_act_backward")
        return dx, None


def activation(x, act):
    return _Act.apply(x, act)
