"""ctypes binding of libmgvae_hip.so (the C ABI in include/mgvae.h).

There is deliberately NO fallback: if the shared library is missing or a tensor is not
on a ROCm device the call raises.  The library is built in-tree by
``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmgvae_hip.so")

c_int, c_float, c_size_t, c_void_p, c_u64 = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint64


class ConvDesc(ctypes.Structure):
    """MgvaeConvDesc (include/mgvae.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "Cx", "H", "W", "Cy", "OH", "OW", "KH", "KW", "SH", "SW", "PH", "PW",
        "x_ctot", "x_coff", "y_ctot", "y_coff", "act")] + [("slope", ctypes.c_float)]


class ActMask(ctypes.Structure):
    """MgvaeActMask (include/mgvae.h)"""
    _fields_ = [("src", ctypes.c_void_p), ("ctot", ctypes.c_int32), ("coff", ctypes.c_int32), ("act", ctypes.c_int32),
                ("slope", ctypes.c_float)]


class ChainCall(ctypes.Structure):
    """MgvaeChainCall (include/mgvae.h)"""
    _fields_ = [("fn", ctypes.c_int32), ("nargs", ctypes.c_int32), ("first", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class ProfRec(ctypes.Structure):  # noqa: E302
    _fields_ = [("kind", ctypes.c_int32), ("tile", ctypes.c_int32), ("launches", ctypes.c_int32),
                ("ms", ctypes.c_double), ("flops", ctypes.c_double)]


P = c_void_p
# name -> (restype, argtypes); every symbol include/mgvae.h declares
SIGNATURES = {
    "mgvae_strerror": (ctypes.c_char_p, [c_int]),
    "mgvae_device_info": (c_int, [ctypes.c_char_p, c_size_t, ctypes.POINTER(c_int)]),
    "mgvae_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P]),
    "mgvae_conv2d_bwd_data": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P]),
    "mgvae_weight_transpose": (c_int, [P, P, c_int, c_int, c_int, P]),
    "mgvae_conv2d_bwd_data_tw": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P]),
    "mgvae_unpack_bits": (c_int, [P, P, ctypes.c_size_t, P]),
    "mgvae_set_compute_dtype": (c_int, [c_int]),
    "mgvae_get_compute_dtype": (c_int, []),
    "mgvae_conv2d_fwd_masked": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P]),
    "mgvae_conv2d_bwd_data_masked": (c_int, [ctypes.POINTER(ConvDesc), P, P, c_int, P, P, ctypes.POINTER(ActMask), P]),
    "mgvae_conv2d_bwd_weight": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P]),
    "mgvae_conv2d_nhwc_fwd": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P]),
    "mgvae_conv2d_nhwc_bwd_data": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P]),
    "mgvae_conv2d_nhwc_bwd_weight": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P]),
    "mgvae_pack_conv_weights_bf16": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "mgvae_conv2d_nhwc_bf16_workspace": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "mgvae_conv2d_nhwc_bf16_fwd": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P, c_size_t, P]),
    "mgvae_conv2d_nhwc_bf16_bwd_data": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P, c_size_t, P]),
    "mgvae_conv2d_nhwc_bf16_bwd_weight": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P]),
    "mgvae_pack_conv_weights_x3": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "mgvae_pack_conv_weights_grouped": (c_int, [P, c_int, c_int, c_int, P]),
    "mgvae_conv2d_nhwc_x3_workspace": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "mgvae_conv2d_nhwc_x3_fwd": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P, c_size_t, P]),
    "mgvae_conv2d_nhwc_x3_bwd_data": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ActMask), P, c_size_t, P]),
    "mgvae_conv2d_nhwc_x3_bwd_weight": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P]),
    "mgvae_conv_pack_floats": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "mgvae_conv_pack": (c_int, [ctypes.POINTER(ConvDesc), c_int, P, P, P]),
    "mgvae_conv2d_fwd_packed": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P]),
    "mgvae_conv2d_bwd_data_packed": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P]),
    "mgvae_channel_sum_accum": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "mgvae_instance_norm_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, P, P, P, P]),
    "mgvae_instance_norm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, P, P, P]),
    "mgvae_batch_norm_fwd": (c_int, [P] * 7 + [c_int] * 4 + [c_float, c_float, c_int, c_float, P]),
    "mgvae_batch_norm_bwd": (c_int, [P] * 8 + [c_int] * 5 + [c_float, P]),
    "mgvae_group_sum_fwd": (c_int, [P, P, c_size_t, c_int, c_int, P]),
    "mgvae_group_sum_bwd": (c_int, [P, P, c_size_t, c_int, c_int, P]),
    "mgvae_cbam_save_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mgvae_cbam_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "mgvae_cbam_bwd_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mgvae_cbam_bwd": (c_int, [P] * 13 + [c_int] * 8 + [c_float, c_int, P]),
    "mgvae_norm_cbam_nhwc_save_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mgvae_norm_cbam_nhwc_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mgvae_norm_cbam_nhwc_fwd": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P] + [c_int] * 6 + [c_float, c_int, c_int, c_float, c_int, P]),
    "mgvae_norm_cbam_nhwc_bwd": (c_int, [P] * 17 + [c_int] * 8 + [c_float, c_int, P]),
    "mgvae_norm_cbam_nhwc_bwd_mlp_wgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "mgvae_instance_norm_nhwc_stats_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mgvae_instance_norm_nhwc_fwd": (c_int, [P, P, P, P, P] + [c_int] * 6 + [c_float, c_int, c_float, c_int, P]),
    "mgvae_instance_norm_nhwc_bwd": (c_int, [P] * 9 + [c_int] * 7 + [c_float, c_int, P]),
    "mgvae_channel_sum_nhwc_accum": (c_int, [P, ctypes.c_long, c_int, c_int, c_int, P, c_int, P]),
    "mgvae_layout_nchw_to_nhwc": (c_int, [P, P] + [c_int] * 8 + [P]),
    "mgvae_layout_nhwc_to_nchw": (c_int, [P, P] + [c_int] * 8 + [P]),
    "mgvae_mean_nhwc_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "mgvae_mean_nhwc_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "mgvae_conv2d_c1_nhwc_fwd": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, c_int, P]),
    "mgvae_conv2d_c1_nhwc_bwd_weight": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, c_int, P]),
    "mgvae_conv2d_to1_nhwc_fwd": (c_int, [P, P, P, ctypes.c_long, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "mgvae_conv2d_to1_nhwc_bwd": (c_int, [P, P, P, P, P, P, ctypes.c_long, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "mgvae_cast_storage": (c_int, [P, c_int, P, c_int, c_size_t, P]),
    "mgvae_act_bwd": (c_int, [P, P, P] + [c_int] * 10 + [c_float, P]),
    "mgvae_copy2d": (c_int, [P, c_size_t, P, c_size_t, c_size_t, c_size_t, P]),
    "mgvae_add_inplace": (c_int, [P, P, c_size_t, P]),
    "mgvae_maxpool2_fwd": (c_int, [P, P, P, c_size_t, c_int, c_int, P]),
    "mgvae_maxpool2_bwd": (c_int, [P, P, P, c_size_t, c_int, c_int, P]),
    "mgvae_act_fwd": (c_int, [P, P, c_size_t, c_int, c_float, P]),
    "mgvae_axpby": (c_int, [P, P, P, c_float, c_float, c_size_t, P]),
    "mgvae_rowmean_fwd": (c_int, [P, P, c_int, c_int, P]),
    "mgvae_rowmean_bwd": (c_int, [P, P, c_int, c_int, P]),
    "mgvae_embedding_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_size_t, P]),
    "mgvae_embedding_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_size_t, P]),
    "mgvae_dropout_fwd": (c_int, [P, P, P, c_size_t, c_float, c_u64, c_u64, P]),
    "mgvae_mul": (c_int, [P, P, P, c_size_t, P]),
    "mgvae_randn": (c_int, [P, c_size_t, c_float, c_u64, c_u64, P]),
    "mgvae_bce_partial_floats": (c_size_t, []),
    "mgvae_bce_fwd": (c_int, [P, P, P, c_float, c_size_t, c_int, c_int, P, P, P]),
    "mgvae_bce_bwd": (c_int, [P, P, P, c_float, c_size_t, c_int, P, P, P]),
    "mgvae_reparam_kl_fwd": (c_int, [P, P, P, P, P, P, c_size_t, P]),
    "mgvae_reparam_kl_bwd": (c_int, [P, P, P, P, P, P, P, c_size_t, P]),
    "mgvae_adam_step": (c_int, [P, P, P, P, c_size_t, P, c_float, c_float, P]),
    "mgvae_f32_to_bf16": (c_int, [P, P, c_size_t, P]),
    "mgvae_bf16_to_f32": (c_int, [P, P, c_size_t, P]),
    "mgvae_bf16_rows_sum": (c_int, [P, P, c_int, c_size_t, P]),
    "mgvae_chain_fn_count": (c_int, []),
    "mgvae_chain_fn_id": (c_int, [ctypes.c_char_p]),
    "mgvae_chain_run": (c_int, [ctypes.POINTER(ChainCall), c_int, P, ctypes.POINTER(c_int)]),
    "mgvae_stream_fork": (c_int, [P, P]),
    "mgvae_add_inplace_typed": (c_int, [P, P, c_size_t, c_int, P]),
    "mgvae_prof_enable": (c_int, [c_int]),
    "mgvae_prof_collect": (c_int, [ctypes.POINTER(ProfRec), c_int]),
    "mgvae_prof_detail": (c_int, [ctypes.c_char_p]),
    "mgvae_prof_record_begin": (c_int, [c_int, c_int, ctypes.c_double, P, ctypes.POINTER(P)]),
    "mgvae_prof_record_end": (c_int, [P, P]),
    "mgvae_kernel_name": (ctypes.c_char_p, [c_int, c_int]),
}

_lib = None


def lib():
    """Load libmgvae_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmgvae_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU / eager fallback for the HIP hot path." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)     # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (what, lib().mgvae_strerror(rc).decode(), rc))
