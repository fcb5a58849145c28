"""ctypes binding of libicamd.so (include/icamd.h) -- the only way the package reaches the GPU kernels.

There is no CPU or PyTorch fallback: if the shared library is missing or a symbol is absent, import of the
compute path raises.  PyTorch is used for device memory (tensors), streams and torch.distributed only.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libicamd.so")

ABI_VERSION = 5
ERRORS = {1: "ICAMD_ERR_BAD_ARG", 2: "ICAMD_ERR_UNSUPPORTED", 3: "ICAMD_ERR_WORKSPACE", 4: "ICAMD_ERR_LAUNCH"}


class IcamdError(RuntimeError):
    pass


class BnBwdFuse(Structure):
    _fields_ = [("y", c_void_p), ("mask_src", c_void_p), ("mean", c_void_p), ("invstd", c_void_p), ("scale", c_void_p),
                ("shift", c_void_p), ("partials", c_void_p), ("relu", c_int)]


class ImageDesc(Structure):
    """include/icamd.h icamd_image_desc (88 bytes)."""
    _fields_ = [("src_offset", ctypes.c_int64), ("src_h", c_int), ("src_w", c_int), ("crop_top", c_int), ("crop_left", c_int),
                ("crop_h", c_int), ("crop_w", c_int), ("hflip", c_int), ("vflip", c_int), ("jitter_order", c_int * 3),
                ("jitter_factor", c_float * 3), ("erase_top", c_int), ("erase_left", c_int), ("erase_h", c_int),
                ("erase_w", c_int), ("erase_seed", ctypes.c_uint32), ("reserved", c_int)]


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in ("N", "IH", "IW", "Cin", "OH", "OW", "Cout", "KH", "KW", "stride", "pad")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def conv_desc(N, IH, IW, Cin, Cout, KH, KW, stride, pad):
    OH = (IH + 2 * pad - KH) // stride + 1
    OW = (IW + 2 * pad - KW) // stride + 1
    return ConvDesc(N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad)


# name -> (restype, argtypes); mirrors include/icamd.h one to one
_P = c_void_p
_SIGNATURES = {
    "icamd_abi_version": (c_int, []),
    "icamd_conv2d_fwd_gelu": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "icamd_conv2d_dgrad_gelu": (c_int, [_P, _P, _P, _P, _P, _P]),
    "icamd_conv2d_fwd_act": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P]),
    "icamd_bn_fold_filters": (c_int, [_P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P]),
    "icamd_conv2d_stats_rows": (c_int, [POINTER(ConvDesc)]),
    "icamd_conv2d_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "icamd_conv2d_dgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "icamd_conv2d_dgrad_sub2": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "icamd_conv2d_dgrad_stats_rows": (c_int, [POINTER(ConvDesc)]),
    "icamd_conv2d_dgrad_bnbwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, POINTER(BnBwdFuse), _P]),
    "icamd_conv2d_dgrad_bnred_supported": (c_int, [POINTER(ConvDesc)]),
    "icamd_conv2d_dgrad_bnred": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "icamd_bn_bwd_from_gy_partials": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P, c_size_t, _P]),
    "icamd_bn_apply_conv1x1_fused_supported": (c_int, [POINTER(ConvDesc)]),
    "icamd_bn_apply_conv1x1_fused": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "icamd_conv1x1_bn_bwd_fused_supported": (c_int, [POINTER(ConvDesc)]),
    "icamd_conv1x1_bn_bwd_fused_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "icamd_conv1x1_bn_bwd_fused": (c_int, [POINTER(ConvDesc), _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_size_t,
                                           _P, c_size_t, _P]),
    "icamd_conv2d_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "icamd_conv2d_wgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P, c_size_t, _P]),
    "icamd_conv2d_wgrad_bias": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "icamd_filter_transpose": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "icamd_filter_transpose_tiled": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "icamd_bn_workspace_bytes": (c_size_t, [c_int]),
    "icamd_bn_train_finalize": (c_int, [_P, c_int, c_int, c_double, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "icamd_bn_eval_coeffs": (c_int, [c_int, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "icamd_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "icamd_bn_apply_res_bn": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "icamd_bn_bwd_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "icamd_bn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, c_int, _P, c_size_t, _P]),
    "icamd_bn_bwd_maxpool3x3s2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "icamd_bn_bwd_dual": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int,
                                  _P, _P, c_size_t, _P]),
    "icamd_bn_bwd_apply_workspace_bytes": (c_size_t, [c_int]),
    "icamd_bn_bwd_from_partials": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P, c_size_t, _P]),
    "icamd_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_float, _P]),
    "icamd_layernorm_bwd_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "icamd_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P, c_size_t, _P]),
    "icamd_gelu_fwd": (c_int, [_P, _P, c_longlong, _P]),
    "icamd_gelu_bwd": (c_int, [_P, _P, _P, c_longlong, _P]),
    "icamd_colsum_rows_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "icamd_colsum_rows": (c_int, [_P, c_longlong, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "icamd_dwconv7_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "icamd_dwconv7_dgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "icamd_dwconv7_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "icamd_dwconv7_wgrad": (c_int, [_P, _P, _P, c_int, _P, c_size_t, c_int, c_int, c_int, c_int, _P]),
    "icamd_dwconv7_wgrad_bias_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "icamd_dwconv7_wgrad_bias": (c_int, [_P, _P, _P, _P, c_int, _P, c_size_t, c_int, c_int, c_int, c_int, _P]),
    "icamd_layerscale_fwd": (c_int, [_P, _P, _P, _P, _P, c_longlong, c_int, c_longlong, _P]),
    "icamd_layerscale_bwd_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "icamd_layerscale_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, c_longlong, c_int, _P, c_size_t, _P]),
    "icamd_layerscale_fold": (c_int, [_P, _P, _P, _P, c_int, c_int, c_longlong, _P]),
    "icamd_rows_fix": (c_int, [_P, c_int, _P, _P, c_longlong, _P, c_longlong, _P]),
    "icamd_dropped_colsum": (c_int, [_P, _P, c_int, c_longlong, c_int, _P, _P]),
    "icamd_layerscale_param_grads": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_float, c_int, c_int, _P, _P, _P, c_int, _P]),
    "icamd_vit_tokens_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "icamd_batch_sum": (c_int, [_P, c_longlong, c_int, c_longlong, _P, c_int, _P]),
    "icamd_strided_rows_copy": (c_int, [_P, c_longlong, _P, c_longlong, c_longlong, c_longlong, _P]),
    "icamd_fill_zero": (c_int, [_P, c_size_t, _P]),
    "icamd_attention_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "icamd_attention_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "icamd_maxpool3x3s2_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "icamd_bn_relu_maxpool3x3s2_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "icamd_maxpool3x3s2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "icamd_avgpool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "icamd_avgpool_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "icamd_pack_input": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, _P]),
    "icamd_pack_input_rgb4": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, _P]),
    "icamd_stem7x7s2_stats_rows": (c_int, [c_int, c_int, c_int]),
    "icamd_stem7x7s2_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "icamd_stem7x7s2_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "icamd_stem7x7s2_wgrad": (c_int, [_P, _P, _P, c_int, _P, c_size_t, c_int, c_int, c_int, c_int, _P]),
    "icamd_image_pipeline_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "icamd_image_pipeline": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), _P,
                                     _P, c_size_t, _P]),
    "icamd_image_pipeline_u8": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "icamd_softmax_xent": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_float, c_float, c_float, _P, _P, _P, _P]),
    "icamd_step_metrics": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "icamd_grad_norm_workspace_bytes": (c_size_t, []),
    "icamd_grad_norm": (c_int, [_P, c_longlong, c_float, c_float, _P, _P, _P]),
    "icamd_adamw_ema": (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_float, c_int,
                                c_float, c_float, _P, _P, _P, c_int, _P]),
    "icamd_grad_guard": (c_int, [_P, c_longlong, _P, _P]),
    "icamd_optim_ema": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_float,
                                c_int, c_float, c_float, _P, _P, _P, c_int, _P]),
    "icamd_lerp": (c_int, [_P, _P, c_longlong, c_float, _P, _P]),
    "icamd_f32_to_bf16": (c_int, [_P, _P, c_longlong, _P]),
    "icamd_colsum": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    "icamd_rccl_available": (c_int, []),
    "icamd_rccl_version": (c_int, []),
    "icamd_rccl_unique_id": (c_int, [_P]),
    "icamd_rccl_comm_init": (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    "icamd_rccl_comm_info": (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    "icamd_rccl_comm_destroy": (c_int, [_P]),
    "icamd_allreduce_bucket_launch": (c_int, [_P, _P, c_longlong, c_int, c_int, _P]),
    "icamd_broadcast_launch": (c_int, [_P, _P, c_longlong, c_int, c_int, _P]),
    "icamd_prof_enable": (c_int, [c_int]),
    "icamd_prof_classes": (c_int, []),
    "icamd_prof_collect": (c_int, [POINTER(c_double), POINTER(c_longlong), POINTER(c_double), POINTER(c_double), c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libicamd.so and bind every symbol of include/icamd.h; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IcamdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.icamd_abi_version() != ABI_VERSION:
        raise IcamdError("libicamd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise IcamdError(f"{what} failed: {ERRORS.get(rc, rc)}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    if not torch.cuda.is_available():
        raise IcamdError("the imageclassification_amd compute path needs an AMD GPU (gfx950); none is visible")


PROF_CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bn_finalize", "bn_apply", "bn_bwd", "pool", "pack", "loss",
                "optimizer", "misc", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "elementwise", "dwconv", "conv_bn_bwd_fused", "bn_apply_conv_fused")


def prof_collect(work=False):
    """{class: (elapsed_ms, calls)} accumulated since the last call (HIP events on the launch stream); with work=True
    {class: (elapsed_ms, calls, algorithmic_bytes, algorithmic_flops)} (include/icamd.h: icamd_prof_collect)."""
    lib = load()
    n = lib.icamd_prof_classes()
    assert n == len(PROF_CLASSES), (n, len(PROF_CLASSES))
    ms = (c_double * n)()
    calls = (c_longlong * n)()
    nbytes = (c_double * n)()
    flops = (c_double * n)()
    check(lib.icamd_prof_collect(ms, calls, nbytes, flops, n), "prof_collect")
    if work:
        return {PROF_CLASSES[i]: (ms[i], calls[i], nbytes[i], flops[i]) for i in range(n)}
    return {PROF_CLASSES[i]: (ms[i], calls[i]) for i in range(n)}
