"""ctypes binding of ``libmcvc_hip.so`` (the C ABI declared in ``include/mcvc.h``).

PyTorch is used only as the owner of device memory and HIP streams: every call below hands raw
device pointers and the current stream to the library.  There is NO fallback -- if the library is
missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p

import torch  # noqa: F401  (must be imported first so libamdhip64.so.7 is already resident)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCVC_LIB: another build of the same library (same-box A/B of two kernel variants: tools/ab_lib.sh); it must export the same ABI version
LIB_PATH = os.environ.get("MCVC_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libmcvc_hip.so")

ABI_VERSION = 3                       # include/mcvc.h MCVC_ABI_VERSION
GEN_NPARAMS = 110
DISC_NPARAMS = 20
N_MEL = 80

_lib = None

_PP = ctypes.POINTER(c_void_p)

_SIGS = {
    "mcvc_version": (c_int, []),
    "mcvc_set_deterministic": (c_int, [c_int]),
    "mcvc_get_deterministic": (c_int, []),
    "mcvc_set_precise": (c_int, [c_int]),
    "mcvc_get_precise": (c_int, []),
    "mcvc_set_trunk_persistent": (c_int, [c_int]),
    "mcvc_twin_begin": (c_int, []),
    "mcvc_twin_switch": (c_int, []),
    "mcvc_twin_end": (c_int, []),
    "mcvc_twin_launches": (c_int, []),
    "mcvc_gen_packed_floats": (c_longlong, []),
    "mcvc_disc_packed_floats": (c_longlong, []),
    "mcvc_gen_stash_floats": (c_longlong, [c_int, c_int]),
    "mcvc_gen_scratch_floats": (c_longlong, [c_int, c_int]),
    "mcvc_disc_stash_floats": (c_longlong, [c_int, c_int]),
    "mcvc_disc_scratch_floats": (c_longlong, [c_int, c_int]),
    "mcvc_gen_out_frames": (c_int, [c_int]),
    "mcvc_disc_out_frames": (c_int, [c_int]),
    "mcvc_gen_pack": (c_int, [_PP, c_void_p, c_void_p]),
    "mcvc_gen_trunk_fused": (c_int, [c_int, c_int]),
    "mcvc_gen_pack_small_batch": (c_int, [_PP, c_void_p, c_int, c_int, c_void_p]),
    "mcvc_gen_pack_sets": (c_int, [_PP, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mcvc_gen_pack_ranges": (c_int, [_PP, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mcvc_disc_pack": (c_int, [_PP, c_void_p, c_void_p]),
    "mcvc_disc_pack_small": (c_int, [_PP, c_void_p, c_int, c_void_p]),
    "mcvc_disc_pack_batch": (c_int, [_PP, c_void_p, c_int, c_int, c_void_p]),
    "mcvc_gen_forward": (c_int, [_PP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "mcvc_gen_backward": (c_int, [_PP, c_void_p, _PP, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "mcvc_gen_backward_overlap": (c_int, [_PP, c_void_p, _PP, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_int, c_int,
                                          c_void_p, c_void_p, _PP]),
    "mcvc_gen_backward_flags": (c_int, [_PP, c_void_p, _PP, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_int, c_int,
                                        c_void_p, c_void_p, _PP, c_int]),
    "mcvc_gen_backward_prefix": (c_int, [_PP, c_void_p, _PP, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_longlong, c_int, c_int,
                                         c_void_p, c_void_p, _PP, c_int]),
    "mcvc_gen_backward_window": (c_int, [_PP, c_void_p, _PP, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_longlong, c_int, c_int,
                                         c_void_p, c_void_p, _PP, c_int]),
    "mcvc_set_trunk_passes_in_flight": (c_int, [c_int]),
    "mcvc_gen_trunk_persistent": (c_int, [c_int, c_int]),
    "mcvc_gen_trunk_fault": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "mcvc_debug_trunk_fault_inject": (c_int, [c_int]),
    "mcvc_gen_bf16_packed_bytes": (c_longlong, []),
    "mcvc_gen_bf16_workspace_bytes": (c_longlong, [c_int, c_int]),
    "mcvc_gen_bf16_pack": (c_int, [_PP, c_void_p, c_void_p]),
    "mcvc_gen_infer_bf16": (c_int, [_PP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "mcvc_bf16_conv2d_pack_bytes": (c_longlong, [c_int, c_int, c_int, c_int]),
    "mcvc_bf16_conv2d": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    "mcvc_bf16_instnorm_act": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "mcvc_bf16_c2d1d_pack_bytes": (c_longlong, []),
    "mcvc_bf16_c2d1d_norm": (c_int, [c_void_p] * 6 + [c_int] * 2 + [c_void_p]),
    "mcvc_bf16_trunk_layer_pack_bytes": (c_longlong, [c_int] * 3),
    "mcvc_bf16_trunk_layer": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p]),
    "mcvc_bf16_last_conv_pack_bytes": (c_longlong, []),
    "mcvc_bf16_last_conv": (c_int, [c_void_p] * 5 + [c_int] * 2 + [c_void_p]),
    "mcvc_bf16_conv1_glu_pack_bytes": (c_longlong, []),
    "mcvc_bf16_conv1_glu": (c_int, [c_void_p] * 8 + [c_int] * 2 + [c_void_p]),
    "mcvc_disc_forward": (c_int, [_PP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "mcvc_disc_backward": (c_int, [_PP, c_void_p, _PP, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "mcvc_l1_loss": (c_int, [c_void_p, c_void_p, c_longlong, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mcvc_lsgan_loss": (c_int, [c_void_p, c_longlong, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcvc_loss_combine": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcvc_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]),
    "mcvc_adam_step2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]),
    "mcvc_gen_update_ranges": (c_int, [_PP, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_float, c_float, c_float, c_float, c_int, c_float, c_int, c_void_p]),
    "mcvc_disc_update_batch": (c_int, [_PP, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_float, c_float, c_float, c_float, c_int, c_float, c_int, c_void_p]),
    "mcvc_draw_batch": (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_int,
                                ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcvc_axpy": (c_int, [c_void_p, c_void_p, c_float, c_longlong, c_void_p]),
    "mcvc_conv2d_pack_floats": (c_longlong, [c_int, c_int, c_int, c_int]),
    "mcvc_conv2d_forward": (c_int, [c_void_p] * 6 + [c_int] * 12 + [c_void_p]),
    "mcvc_conv2d_dgrad": (c_int, [c_void_p] * 5 + [c_int] * 11 + [c_void_p]),
    "mcvc_conv2d_wgrad_slab_floats": (c_longlong, [c_int] * 10),
    "mcvc_conv2d_wgrad": (c_int, [c_void_p] * 4 + [c_longlong] + [c_int] * 10 + [c_void_p]),
    "mcvc_layer_packed_floats": (c_longlong, [c_int] * 8),
    "mcvc_layer_scratch_floats": (c_longlong, [c_int] * 11),
    "mcvc_layer_pack": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "mcvc_layer_forward": (c_int, [c_void_p] * 6 + [c_longlong] + [c_int] * 13 + [c_void_p]),
    "mcvc_layer_dgrad": (c_int, [c_void_p] * 6 + [c_longlong] + [c_int] * 12 + [c_void_p]),
    "mcvc_layer_wgrad": (c_int, [c_void_p] * 5 + [c_longlong] + [c_int] * 12 + [c_void_p]),
    "mcvc_trunk_layer_backward": (c_int, [c_void_p] * 19 + [c_int] * 5 + [c_void_p]),
    "mcvc_instnorm_act_forward": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_void_p]),
    "mcvc_instnorm_act_backward": (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_void_p]),
    "mcvc_trunk_layer_forward": (c_int, [c_void_p] * 13 + [c_int] * 5 + [c_void_p]),
    "mcvc_batched_gemm": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_longlong] * 3 + [c_void_p]),
    "mcvc_bias_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mcvc_act_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mcvc_act_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mcvc_fif_input": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mcvc_fif_input_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mcvc_trace_enable": (c_int, [c_int]),
    "mcvc_trace_kinds": (c_int, []),
    "mcvc_trace_kind_name": (ctypes.c_char_p, [c_int]),
    "mcvc_trace_collect": (c_int, [ctypes.POINTER(ctypes.c_double)]),
    "mcvc_trace_collect_raw": (c_int, [ctypes.POINTER(ctypes.c_double), c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    """Load (once) and return the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmcvc_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the ABI and the header ever diverge
            fn.restype = res
            fn.argtypes = args
        if L.mcvc_version() != ABI_VERSION:
            raise RuntimeError("libmcvc_hip.so ABI version mismatch")
        _lib = L
    return _lib


class twin(object):
    """``with twin() as tw: <calls on network 0>; tw.switch(); <the same calls on network 1>`` -- the two sequences go out as ONE set of
    grouped launches (include/mcvc.h, mcvc_twin_*).  An exception inside the block abandons the bracket."""

    def __enter__(self):
        check(lib().mcvc_twin_begin(), "mcvc_twin_begin")
        return self

    def switch(self):
        check(lib().mcvc_twin_switch(), "mcvc_twin_switch")

    def __exit__(self, et, ev, tb):
        rc = lib().mcvc_twin_end()
        if et is None:
            check(rc, "mcvc_twin_end (the two call sequences of a grouped pass differ)")
        return False


def check(rc: int, what: str = "mcvc call"):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_table(tensors):
    """Host array of device pointers (NULL for None entries)."""
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def require_cuda_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mask_cyclegan_vc (MI355X build): tensors must live on a HIP device; "
                               "there is no CPU path in this package")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("expected contiguous float32 tensors")
