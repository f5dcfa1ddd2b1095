"""ctypes binding of libsrlz_hip.so (the C ABI declared in include/srlz.h).

This is the stub a maintainer of the reference would add: every entry point takes raw device
pointers (``tensor.data_ptr()``), plain ints/floats, POD descriptors and the HIP stream of the caller.
There is NO fallback: if the library is missing the import fails loudly, and every non-zero status raises.
"""
import ctypes
import os
from ctypes import c_int, c_longlong, c_float, c_double, c_void_p, c_size_t, c_char_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrlz_hip.so")


ABI_VERSION = 104  # include/srlz.h SRLZ_ABI_VERSION these prototypes were written for


class SrlzError(RuntimeError):
    pass


class Conv64Desc(Structure):
    _fields_ = [("n", c_int), ("hi", c_int), ("wi", c_int), ("ho", c_int), ("wo", c_int), ("ksize", c_int),
                ("stride", c_int), ("pad", c_int), ("transposed", c_int), ("groups", c_int)]


class SkinnyDesc(Structure):
    _fields_ = [("n", c_int), ("c", c_int), ("himg", c_int), ("wimg", c_int), ("hf", c_int), ("wf", c_int),
                ("kind", c_int), ("groups", c_int)]


class PoolDesc(Structure):
    _fields_ = [("n", c_int), ("h", c_int), ("w", c_int), ("hp", c_int), ("wp", c_int), ("pool_pad", c_int),
                ("out_nchw", c_int), ("groups", c_int)]


class ConvNDesc(Structure):
    _fields_ = [("n", c_int), ("hi", c_int), ("wi", c_int), ("ho", c_int), ("wo", c_int), ("cin", c_int), ("cout", c_int),
                ("ksize", c_int), ("stride", c_int), ("pad", c_int), ("groups", c_int)]

    def __init__(self, n, hi, wi, ho, wo, cin, cout, ksize, stride, pad, groups=1):
        super(ConvNDesc, self).__init__(n, hi, wi, ho, wo, cin, cout, ksize, stride, pad, groups)


class BnReplayItem(Structure):
    """srlz_bn_replay_item"""
    _fields_ = [("batch_stat", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p)]


class BnBwdOperand(Structure):
    """srlz_bn_bwd_operand: raw device pointers + count; build with bn_bwd_operand() so the tensors stay referenced."""
    _fields_ = [("y", c_void_p), ("bnp", c_void_p), ("sums", c_void_p), ("count", c_longlong), ("training", c_int),
                ("dy_out", c_void_p)]


P = c_void_p
_BO = POINTER(BnBwdOperand)
_C64 = POINTER(Conv64Desc)
_SK = POINTER(SkinnyDesc)
_PD = POINTER(PoolDesc)
_CN = POINTER(ConvNDesc)

# name -> (restype, argtypes); restype c_int functions are status-checked
_PROTOS = {
    "srlz_version": (c_int, []),
    "srlz_last_error": (c_char_p, []),
    "srlz_device_cus": (c_int, []),
    "srlz_conv64_packed_floats": (c_size_t, []),
    "srlz_conv64_pack_weights": (c_int, [P, P, P, _C64, P]),
    "srlz_conv64_fwd_tiles": (c_int, [_C64]),
    "srlz_conv64_fwd": (c_int, [P, P, P, P, P, P, _C64, P]),
    "srlz_conv64_bwd_data": (c_int, [P, P, P, _BO, _C64, P]),
    "srlz_conv64_bwd_weight_workspace": (c_size_t, [_C64]),
    "srlz_conv64_bwd_fused_supported": (c_int, [_C64]),
    "srlz_conv64_gather_pipe_supported": (c_int, [_C64, c_int]),
    "srlz_conv64_bwd_fused_workspace": (c_size_t, [_C64]),
    "srlz_conv64_bwd_fused": (c_int, [P, P, P, _BO, P, P, P, P, P, P, c_size_t, _C64, P]),
    "srlz_conv64_bwd_fused_bn_rows": (c_int, [_C64]),
    "srlz_conv64_wino_supported": (c_int, [_C64]),
    "srlz_conv64_wino_packed_floats": (c_size_t, []),
    "srlz_conv64_wino_pack_weights": (c_int, [P, P, P, P]),
    "srlz_conv64_wino_tiles": (c_int, [_C64]),
    "srlz_conv64_wino_fwd": (c_int, [P, P, P, P, P, P, _C64, P]),
    "srlz_conv64_wino_bwd_data": (c_int, [P, P, P, _C64, P]),
    "srlz_conv64_wino_bwd_weight_workspace": (c_size_t, [_C64]),
    "srlz_conv64_wino_bwd_weight": (c_int, [P, P, P, P, c_size_t, _C64, P]),
    "srlz_conv64_wino_bwd_data_rows": (c_int, [_C64]),
    "srlz_conv64_wino_bwd_data_pool_sums": (c_int, [P, P, P, P, P, P, P, _PD, P, _C64, P]),
    "srlz_conv64_bwd_weight": (c_int, [P, P, P, P, P, _BO, P, c_size_t, _C64, P]),
    "srlz_conv64_debug_program": (c_int, [_C64, c_int, POINTER(c_int), c_int]),
    "srlz_debug_placement": (c_int, [P, c_int, c_int, c_int, P]),
    "srlz_debug_mfma_peak": (c_int, [P, c_int, c_int, P]),
    "srlz_debug_mfma_valu": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P]),
    "srlz_convn_packed_floats": (c_size_t, [_CN]),
    "srlz_convn_pack_weights": (c_int, [P, P, _CN, P]),
    "srlz_convn_fwd_tiles": (c_int, [_CN]),
    "srlz_convn_fwd": (c_int, [P, P, P, P, P, _CN, P]),
    "srlz_bn_finalize_chunks_workspace": (c_size_t, [c_int, c_int]),
    "srlz_bn_finalize_chunks": (c_int, [P, c_int, c_int, c_int, c_longlong, P, P, c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "srlz_bn_eval_params_chunks": (c_int, [P, P, P, P, c_float, c_int, P, P]),
    "srlz_bn_add_relu": (c_int, [P, P, P, P, P, c_longlong, c_int, c_int, P]),
    "srlz_avgpool_nhwc": (c_int, [P, P, c_int, c_int, c_int, P]),
    "srlz_prelu_fwd": (c_int, [P, P, P, c_longlong, P]),
    "srlz_prelu_bwd": (c_int, [P, P, P, P, P, c_longlong, P]),
    "srlz_triplet_fwd": (c_int, [P, P, P, c_int, c_int, c_float, P, P, P]),
    "srlz_triplet_bwd": (c_int, [P, P, P, P, P, c_int, c_int, P, P, P, P]),
    "srlz_skinny_tiles": (c_int, [_SK]),
    "srlz_conv64_bwd_data_tiles": (c_int, [_C64]),
    "srlz_conv64_bwd_data_pool_sums": (c_int, [P, P, P, P, P, P, P, _PD, P, _C64, P]),
    "srlz_conv1_fwd": (c_int, [P, P, P, P, _SK, P]),
    "srlz_conv1_fwd_u8": (c_int, [P, P, P, P, P, _SK, P]),
    "srlz_conv1_bwd_weight_fused_u8": (c_int, [P, P, P, P, P, P, P, c_int, P, P, c_size_t, _SK, _PD, P]),
    "srlz_convT_out_fwd_loss_u8": (c_int, [P, P, P, P, P, P, P, P, P, _SK, P]),
    "srlz_normalize_lut": (c_int, [P, P]),
    "srlz_normalize_u8_planar": (c_int, [P, P, P, c_int, c_int, c_longlong, P]),
    "srlz_copy_frames_u8": (c_int, [P, P, c_longlong, P, P, c_longlong, c_int, c_longlong, P]),
    "srlz_copy_frames_u8_strided": (c_int, [P, P, c_longlong, c_longlong, c_longlong, P, P, c_longlong, c_longlong, c_longlong, c_int,
                                            c_longlong, P]),
    "srlz_occlude_frames_u8": (c_int, [P, P, c_longlong, P, P, P, c_int, c_int, c_int, c_int, P]),
    "srlz_skinny_bwd_weight_workspace": (c_size_t, [_SK]),
    "srlz_conv1_bwd_weight": (c_int, [P, P, P, P, c_size_t, _SK, P]),
    "srlz_conv1_bwd_data": (c_int, [P, P, P, _SK, P]),
    "srlz_conv1_bwd_weight_fused": (c_int, [P, P, P, P, P, P, c_int, P, P, c_size_t, _SK, _PD, P]),
    "srlz_bn_relu_pool_bwd_sums": (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, _PD, P]),
    "srlz_convT_out_fwd": (c_int, [P, P, P, P, P, _SK, P]),
    "srlz_convT_out_bwd_data": (c_int, [P, P, P, P, P, P, _SK, P]),
    "srlz_bn_bwd_finalize_partials": (c_int, [P, c_int, c_int, P, P, P, P, c_size_t, P]),
    "srlz_convT_out_bwd_weight": (c_int, [P, P, P, P, P, P, c_size_t, _SK, P]),
    "srlz_convT_out_bwd_fused_supported": (c_int, [_SK]),
    "srlz_convT_out_bwd_fused_tiles": (c_int, [_SK]),
    "srlz_convT_out_bwd_fused_workspace": (c_size_t, [_SK]),
    "srlz_convT_out_bwd_fused": (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, P, c_float, c_float, _SK, P]),
    "srlz_convT_out_fwd_loss_workgroups": (c_int, [_SK]),
    "srlz_convT_out_fwd_loss": (c_int, [P, P, P, P, P, P, P, P, _SK, P]),
    "srlz_pair_loss_finalize": (c_int, [P, c_int, c_longlong, c_int, P, P, P]),
    "srlz_scale_by_scalar": (c_int, [P, P, c_float, c_float, P, c_longlong, P]),
    "srlz_bn_finalize": (c_int, [P, c_int, c_int, c_longlong, P, P, c_float, c_float, c_int, P, P, P, P, P, P, c_size_t, P]),
    "srlz_bn_eval_params": (c_int, [P, P, P, P, c_float, P, P]),
    "srlz_bn_replay": (c_int, [P, c_float, P, P, P]),
    "srlz_bn_replay_many": (c_int, [POINTER(BnReplayItem), c_int, c_float, P]),
    "srlz_bn_relu_pool_fwd": (c_int, [P, P, P, P, _PD, P]),
    "srlz_bn_bwd_workspace": (c_size_t, [c_longlong]),
    "srlz_bn_relu_pool_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, P, c_size_t, _PD, P]),
    "srlz_bn_relu_pool_bwd_apply": (c_int, [P, P, P, P, P, P, c_int, _PD, P]),
    "srlz_bn_relu_fwd": (c_int, [P, P, P, c_longlong, P]),
    "srlz_bn_relu_bwd_sums": (c_int, [P, P, P, P, P, P, P, c_size_t, c_longlong, c_int, P]),
    "srlz_bn_relu_bwd": (c_int, [P, P, P, P, P, P, c_int, P, c_size_t, c_longlong, c_int, P]),
    "srlz_nchw_to_nhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srlz_nhwc_to_nchw": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srlz_linear_workspace": (c_size_t, [c_int, c_int, c_int]),
    "srlz_linear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "srlz_linear_bwd_data": (c_int, [P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    "srlz_linear_fwd_res": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "srlz_linear_bwd_data_res": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "srlz_cat_cols": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "srlz_split_cols": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "srlz_sum_terms": (c_int, [P, c_int, P, c_longlong, P]),
    "srlz_sqdiff_mean": (c_int, [P, P, c_longlong, c_float, P, P, c_size_t, P]),
    "srlz_linear_bwd_weight": (c_int, [P, P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    "srlz_relu_bwd_inplace": (c_int, [P, P, c_longlong, P]),
    "srlz_reduce_workspace": (c_size_t, [c_longlong]),
    "srlz_sqdiff_sum": (c_int, [P, P, c_longlong, P, P, c_size_t, P]),
    "srlz_sqdiff_grad": (c_int, [P, P, P, c_float, P, c_longlong, P]),
    "srlz_sqdiff_sum_groups": (c_int, [P, P, c_longlong, c_int, P, P, c_size_t, P]),
    "srlz_sqdiff_grad_groups": (c_int, [P, P, P, c_int, c_float, c_float, P, c_longlong, c_int, P]),
    "srlz_sqdiff_pair_loss": (c_int, [P, P, c_longlong, c_int, P, P, P, c_size_t, P]),
    "srlz_join2": (c_int, [P, P, P, c_longlong, P]),
    "srlz_weighted_total": (c_int, [P, P, c_int, P, P, P]),
    "srlz_weighted_total_bwd": (c_int, [P, P, c_int, P, P]),
    "srlz_kl_sum": (c_int, [P, P, c_longlong, P, P, c_size_t, P]),
    "srlz_kl_grad": (c_int, [P, P, P, c_float, P, P, c_longlong, P]),
    "srlz_reparam_fwd": (c_int, [P, P, P, P, c_longlong, P]),
    "srlz_reparam_bwd": (c_int, [P, P, P, P, P, c_longlong, P]),
    "srlz_cross_entropy": (c_int, [P, P, c_int, c_int, P, P, P]),
    "srlz_concat_onehot": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "srlz_normalize_u8": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srlz_mask_columns": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srlz_param_norms": (c_int, [P, P, c_int, c_int, c_float, P, P, P]),
    "srlz_param_norms_grad": (c_int, [P, P, P, c_int, c_int, P, P, c_float, P]),
    "srlz_fold_grads": (c_int, [P, P, c_longlong, c_int, P]),
    "srlz_comm_unique_id_bytes": (c_size_t, []),
    "srlz_comm_unique_id": (c_int, [P]),
    "srlz_comm_init": (c_int, [P, c_int, c_int]),
    "srlz_comm_world": (c_int, []),
    "srlz_comm_allreduce_f32": (c_int, [P, c_longlong, P]),
    "srlz_comm_destroy": (c_int, []),
    "srlz_adam_step": (c_int, [P, P, P, P, c_longlong, c_double, c_double, c_double, c_double, c_int, c_float, P]),
    "srlz_adam_step_dev": (c_int, [P, P, P, P, c_longlong, c_double, c_double, c_double, c_double, P, P, c_float, P]),
}

# entry points whose int return value is data, not a status
_NOT_STATUS = {"srlz_version", "srlz_device_cus", "srlz_conv64_fwd_tiles", "srlz_skinny_tiles", "srlz_convn_fwd_tiles",
               "srlz_convT_out_bwd_fused_tiles", "srlz_convT_out_fwd_loss_workgroups", "srlz_conv64_bwd_fused_supported", "srlz_conv64_gather_pipe_supported",
               "srlz_convT_out_bwd_fused_supported",
               "srlz_conv64_bwd_data_tiles", "srlz_conv64_bwd_fused_bn_rows", "srlz_conv64_wino_supported", "srlz_conv64_wino_packed_floats",
               "srlz_conv64_wino_tiles", "srlz_conv64_wino_bwd_data_rows",
               "srlz_conv64_debug_program", "srlz_comm_world"}

EXPORTED = sorted(_PROTOS.keys())


def _load():
    if not os.path.exists(LIB_PATH):
        raise SrlzError(
            "libsrlz_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C srl-zoo_amd/csrc`). There is no CPU fallback for the srl-zoo_amd hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.srlz_version()
    if got != ABI_VERSION:  # a stale build next to newer bindings (or the reverse) would pass arguments in the wrong slots
        raise SrlzError("libsrlz_hip.so at %s reports ABI version %d, these bindings are written for %d (include/srlz.h "
                        "SRLZ_ABI_VERSION): rebuild with `make -C srl-zoo_amd/csrc`" % (LIB_PATH, got, ABI_VERSION))
    return lib


_lib = _load()


def error_text():
    return _lib.srlz_last_error().decode("utf-8", "replace")


_DEBUG_SYNC = bool(os.environ.get("SRLZ_SYNC"))  # debugging aid: device-synchronise after every call


def _wrap(name):
    fn = getattr(_lib, name)
    if _PROTOS[name][0] is c_int and name not in _NOT_STATUS:
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise SrlzError("%s failed (%d): %s" % (name, rc, error_text()))
            if _DEBUG_SYNC:
                import torch
                torch.cuda.synchronize()
            return rc
        call.__name__ = name
        return call
    return fn


for _n in _PROTOS:
    globals()[_n[len("srlz_"):]] = _wrap(_n)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
