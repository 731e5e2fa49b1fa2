"""ctypes binding of libdeva_hip.so (C ABI declared in include/deva_hip.h).

The library is the only compute backend of this package: there is no PyTorch/CPU fallback.
`lib()` raises if the shared object has not been built (`python __graft_entry__.py`), and every
wrapper in `ops.py` raises if handed a tensor that is not a contiguous fp32 HIP tensor.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdeva_hip.so')
ABI_VERSION = 9

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SQUARE_PLUS_ONE = 0, 1, 2, 3
KLAYOUT_TAP_MAJOR, KLAYOUT_CHUNK32 = 0, 1
KLAYOUT_Q4 = 16  # flag: k-quad interleaved storage


class ConvDesc(Structure):
    """mirror of `struct deva_conv_desc` (include/deva_hip.h)"""
    _fields_ = [
        ('in0', c_void_p), ('in1', c_void_p),
        ('in0_batch_stride', c_int64), ('in1_batch_stride', c_int64),
        ('c0', c_int32), ('c1', c_int32),
        ('batch', c_int32), ('height', c_int32), ('width', c_int32),
        ('weight', c_void_p), ('bias', c_void_p),
        ('cout', c_int32), ('cout_pad', c_int32),
        ('k_layout', c_int32),
        ('kh', c_int32), ('kw', c_int32), ('stride', c_int32), ('pad', c_int32),
        ('relu_in', c_int32),
        ('residual', c_void_p), ('residual_batch_stride', c_int64),
        ('act', c_int32),
        ('out', c_void_p),
        ('in_guard_elems', c_int32),
        ('workspace', c_void_p), ('workspace_elems', c_int64),
        ('weight_f16', c_void_p), ('amp', c_int32),
        ('split_scale_log2', c_int32), ('split_flag', c_void_p),
        ('weight_wino', c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol of include/deva_hip.h
SIGNATURES = {
    'deva_hip_version': (c_int, []),
    'deva_hip_last_error': (c_char_p, []),
    'deva_conv2d': (c_int, [POINTER(ConvDesc), c_void_p]),
    'deva_conv_pack_f16': (c_int64, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    'deva_conv_pack_split': (c_int64, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'deva_conv_pack_wino': (c_int64, [c_void_p, c_void_p, c_int, c_int]),
    'deva_conv_pack': (c_int64, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'deva_maxpool3x3s2': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'deva_upsample2x_add': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'deva_area_downsample': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'deva_aggregate': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_void_p]),
    'deva_softmax_channels': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    'deva_upsample4x_softmax': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'deva_global_avgmax': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'deva_cbam_mlp': (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    'deva_cbam_channel_pool': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'deva_cbam_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'deva_stem7x7': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                             c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'deva_stem_pack': (c_int64, [c_void_p, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    'deva_pad2d': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'deva_usage_init': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'deva_upsample2x_add_ds2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'deva_gather_s2': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'deva_gru_update': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'deva_affinity_topk': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_void_p, c_void_p]),
    'deva_affinity_finalize': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'deva_affinity_workspace': (c_int64, [c_int, c_int, c_int]),
    'deva_affinity_force_shape': (c_int, [c_int]),
    'deva_affinity_default_splits': (c_int, [c_int, c_int]),
    'deva_usage_update': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    'deva_readout_sparse': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                    c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'deva_affinity_select': (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    'deva_affinity_read': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'deva_affinity_read_prepared': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int,
                                            c_void_p]),
    'deva_affinity_bank_prep_bytes': (c_int64, [c_int]),
    'deva_affinity_read_scratch': (c_int64, [c_int, c_int, c_int]),
    'deva_affinity_dense': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p]),
    'deva_affinity_prefilter_enabled': (c_int, [c_int, c_int, c_int]),
    'deva_affinity_force_prefilter': (c_int, [c_int]),
    'deva_probe_mfma_f32': (c_int64, [c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    'deva_affinity_read_flag': (c_int, [c_void_p, c_void_p]),
    'deva_affinity_read_stats': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int64), c_void_p]),
    'deva_affinity_merge': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'deva_bank_append': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'deva_bank_gather_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'deva_bank_export': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'deva_rank': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'deva_rank_select': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'deva_evict_select': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'deva_similarity_dense': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                      c_void_p]),
    'deva_softmax_columns': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'deva_label_histogram': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    'deva_lut_remap': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    'deva_input_head': (c_int, [c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int, c_void_p, c_int, c_int,
                                c_int, c_int, c_int, c_int, c_void_p]),
    'deva_index_mask': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'deva_merge_paint': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int64, c_void_p, c_void_p]),
}

_LIB = None


class DevaHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the native library; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DevaHipError(
                f'{LIB_PATH} not found: the HIP kernels are not built. Run `python __graft_entry__.py` '
                '(or `make -C tracking-anything-with-deva_amd/csrc`). There is no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.deva_hip_version() != ABI_VERSION:
            raise DevaHipError(f'libdeva_hip.so ABI {handle.deva_hip_version()} != binding {ABI_VERSION}')
        _LIB = handle
    return _LIB


def check(code: int, what: str) -> None:
    if code != 0:
        raise DevaHipError(f'{what} failed: {lib().deva_hip_last_error().decode()}')
