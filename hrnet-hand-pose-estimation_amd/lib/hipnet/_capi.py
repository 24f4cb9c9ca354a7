"""ctypes binding of libhrnet_hip.so (the C ABI declared in include/hrnet_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a
RuntimeError is raised. The product path never computes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.environ.get('HRNET_HIP_LIB', os.path.join(_PKG, 'csrc', 'libhrnet_hip.so'))

HR_F32, HR_BF16 = 0, 1

(OP_CONV, OP_WGRAD, OP_WGRAD_REDUCE, OP_BN_FINALIZE, OP_SUM_TERMS, OP_GRAD_TERM, OP_BN_BWD_REDUCE,
 OP_BN_BWD_FINALIZE, OP_BILINEAR_CAT, OP_BILINEAR_CAT_BWD, OP_IM2COL_STEM, OP_NHWC_TO_NCHW,
 OP_NCHW_TO_NHWC, OP_PACK_WEIGHTS, OP_BIAS_GRAD, OP_FILL) = range(1, 17)


class HrOp(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('i', ctypes.c_int32 * 19), ('f', ctypes.c_float * 4),
                ('p', ctypes.c_void_p * 14)]


OP_PACK_TABLE, OP_EVENT_RECORD, OP_STREAM_WAIT, OP_WGRAD_REDUCE_TABLE, OP_BWD_FUSED, OP_BN_FINALIZE_TABLE = 17, 18, 19, 20, 21, 22
OP_BWD_PW = 23
OP_CONV_SUM = 24
OP_EW_TABLE = 25
OP_HEAD_MIX = 26
OP_UPSAMPLE_T = 27
OP_HEAD_BWD = 28
OP_POOL_REDUCE = 29
LANE_SLOT = 18
ABI_VERSION = 2      # hrnet_abi_version() of the library this file binds (HrOp slot meanings, table structs)


class HrPackEnt(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p), ('Cout', ctypes.c_int32), ('Cin', ctypes.c_int32),
                ('ks', ctypes.c_int32), ('Cout_pad', ctypes.c_int32), ('Cin_pad', ctypes.c_int32),
                ('mode', ctypes.c_int32), ('block0', ctypes.c_int32), ('ld', ctypes.c_int32)]


class HrWredEnt(ctypes.Structure):
    _fields_ = [('slabs', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('nsplit', ctypes.c_int32),
                ('Cout_pad', ctypes.c_int32), ('Cin_pad', ctypes.c_int32), ('ks', ctypes.c_int32),
                ('Cout', ctypes.c_int32), ('Cin', ctypes.c_int32), ('kflat', ctypes.c_int32),
                ('accumulate', ctypes.c_int32), ('block0', ctypes.c_int32), ('ld', ctypes.c_int32)]


class HrBnEnt(ctypes.Structure):
    _fields_ = [('sums', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p),
                ('running_mean', ctypes.c_void_p), ('running_var', ctypes.c_void_p),
                ('num_batches_tracked', ctypes.c_void_p), ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p),
                ('mean', ctypes.c_void_p), ('invstd', ctypes.c_void_p), ('count', ctypes.c_float),
                ('momentum', ctypes.c_float), ('eps', ctypes.c_float), ('C', ctypes.c_int32), ('block0', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


class HrBnBwdRef(ctypes.Structure):
    _fields_ = [('rows', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('save_mean', ctypes.c_void_p),
                ('save_invstd', ctypes.c_void_p), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p),
                ('count', ctypes.c_float), ('nrows', ctypes.c_int32), ('accumulate', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


_c_int, _c_float, _c_vp, _c_i64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64
_pp = ctypes.POINTER(ctypes.c_void_p)
_ip = ctypes.POINTER(ctypes.c_int)

# name -> argument ctypes (all return int unless listed in _RET)
_SIGS = {
    'hrnet_abi_version': [],
    'hrnet_program_run': [ctypes.POINTER(HrOp), _c_int, _c_vp],
    'hrnet_program_run_streams': [ctypes.POINTER(HrOp), _c_int, _pp, _c_int],
    'hrnet_program_run_streams_timed': [ctypes.POINTER(HrOp), _c_int, _pp, _c_int, ctypes.POINTER(ctypes.c_float)],
    'hrnet_event_destroy': [_c_vp],
    'hrnet_conv2d': [_c_int] + [_c_vp] * 7 + [_c_int] * 12 + [_c_vp],
    'hrnet_conv2d_bwdstats': [_c_int] + [_c_vp] * 8 + [_c_int] * 11 + [_c_vp],
    'hrnet_conv2d_bnref': [_c_int] + [_c_vp] * 5 + [_c_float, _c_float] + [_c_vp] * 3 + [_c_int] * 10 + [_c_vp],
    'hrnet_bn_finalize_table': [_c_vp, _c_int, _c_int, _c_vp],
    'hrnet_lincomb_f32': [_c_vp, _c_i64, _c_int, _pp, ctypes.POINTER(ctypes.c_float), _c_vp],
    'hrnet_conv2d_dilated3x3': [_c_int, _c_vp, _c_vp, _c_i64, _c_vp] + [_c_int] * 6 + [_c_vp],
    'hrnet_conv2d_sum': [_c_int] + [_c_vp] * 8 + [_c_float, _c_float] + [_c_vp] * 3 + [_c_int] * 7 + [_c_vp],
    'hrnet_sum_terms_bnref': [_c_int, _c_vp] + [_c_int] * 5 + [_pp, _pp, _pp, _ip, _ip, _c_int, _c_int, ctypes.POINTER(ctypes.c_float), _c_float, _c_vp],
    'hrnet_conv_mode': [_c_int] * 7,
    'hrnet_ew_table_blocks': [_c_int] * 6,
    'hrnet_conv_ring_enable': [_c_int],
    'hrnet_conv_ring_supported': [_c_int] * 6,
    'hrnet_conv_rows_bwdstats': [_c_int] * 8,
    'hrnet_conv_route': [_c_int] * 8,
    'hrnet_conv_ring_sum_enable': [_c_int],
    'hrnet_conv_tiles': [_c_int] * 6,
    'hrnet_conv_tiles_bwdstats': [_c_int] * 6,
    'hrnet_conv_tile_walk': [_c_int] * 8 + [_ip],
    'hrnet_conv_kernel_name': [_c_int] * 10 + [ctypes.c_char_p, _c_int],
    'hrnet_wgrad_kernel_name': [_c_int] * 7 + [ctypes.c_char_p, _c_int],
    'hrnet_conv2d_wgrad': [_c_int] + [_c_vp] * 5 + [_c_int] * 11 + [_c_vp],
    'hrnet_conv3x3_bwd_fused': [_c_int] + [_c_vp] * 6 + [_c_int] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 3 + [_c_int] * 5 + [_c_vp],
    'hrnet_conv3x3_bwd_fused_bnref': [_c_int] + [_c_vp] * 7 + [_c_int] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 3 + [_c_int] * 5 + [_c_vp],
    'hrnet_conv1x1_bwd_fused_bnref': [_c_int] + [_c_vp] * 7 + [_c_int] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 3 + [_c_i64, _c_int, _c_int] + [_c_vp],
    'hrnet_bwd_fused_supported': [_c_int] * 3,
    'hrnet_bwd_fused_splits': [_c_int] * 6,
    'hrnet_bwd_fused_kernel_name': [_c_int] * 3 + [ctypes.c_char_p, _c_int],
    'hrnet_conv1x1_bwd_fused': [_c_int] + [_c_vp] * 6 + [_c_int] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 3 + [_c_i64, _c_int, _c_int] + [_c_vp],
    'hrnet_bwd_pw_supported': [_c_int] * 3,
    'hrnet_pack_blocks': [_c_int] * 4,
    'hrnet_bwd_pw_rows_supported': [_c_int] * 3,
    'hrnet_bwd_pw_splits': [_c_int, _c_i64, _c_int, _c_int],
    'hrnet_bwd_pw_kernel_name': [_c_int] * 3 + [ctypes.c_char_p, _c_int],
    'hrnet_wgrad_splits': [_c_int] * 8,
    'hrnet_wgrad_tiles': [_c_int] * 8,
    'hrnet_wgrad_blocks_per_split': [_c_int] * 7,
    'hrnet_wgrad_reduce': [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp],
    'hrnet_pack_weights': [_c_int, _c_vp, _c_vp] + [_c_int] * 6 + [_c_vp],
    'hrnet_pack_weights_table': [_c_int, _c_vp, _c_int, _c_int, _c_vp],
    'hrnet_wgrad_reduce_table': [_c_vp, _c_int, _c_int, _c_vp],
    'hrnet_bn_finalize': [_c_vp, _c_int, _c_int, _c_float] + [_c_vp] * 5 + [_c_float, _c_float, _c_int]
                         + [_c_vp] * 4 + [_c_vp],
    'hrnet_sum_terms': [_c_int, _c_vp] + [_c_int] * 5 + [_pp, _pp, _pp, _ip, _ip, _c_int, _c_vp],
    'hrnet_grad_term': [_c_int] + [_c_vp] * 7 + [_c_int] * 7 + [_c_vp],
    'hrnet_grad_term2': [_c_int] + [_c_vp] * 8 + [_c_int] * 6 + [_c_vp],
    'hrnet_bn_bwd_reduce': [_c_int] + [_c_vp] * 6 + [_c_int] * 6 + [_c_vp],
    'hrnet_reduce_blocks': [_c_int] * 4,
    'hrnet_bn_bwd_finalize': [_c_vp, _c_int, _c_int, _c_float] + [_c_vp] * 6 + [_c_int, _c_vp],
    'hrnet_bilinear_cat': [_c_int, _c_vp, _pp, _ip, _ip, _ip] + [_c_int] * 5 + [_c_vp],
    'hrnet_bilinear_cat_bwd': [_c_int, _c_vp, _pp, _ip, _ip, _ip] + [_c_int] * 6 + [_c_vp],
    'hrnet_head_mix': [_c_int] + [_c_vp] * 5 + [_c_int, _pp, _ip, _ip] + [_c_int] * 7 + [_c_vp],
    'hrnet_head_bwd': [_c_int, _c_int] + [_c_vp] * 7 + [_c_int] * 6 + [_c_vp],
    'hrnet_head_mix_rows': [_c_int] * 3,
    'hrnet_head_mix_supported': [_c_int] * 3,
    'hrnet_upsample_bilinear_t': [_c_int, _c_vp, _pp, _ip, _ip] + [_c_int] * 7 + [_c_vp],
    'hrnet_gaussian_targets': [_c_vp] * 3 + [_c_int] * 3 + [_c_float, _c_vp],
    'hrnet_normalize_u8': [_c_vp, _c_vp] + [_c_int] * 3 + [ctypes.POINTER(ctypes.c_float)] * 2 + [_c_vp],
    'hrnet_spatial_softmax_fwd': [_c_vp] * 3 + [_c_int] * 2 + [_c_vp],
    'hrnet_spatial_softmax_bwd': [_c_vp] * 6 + [_c_int] * 2 + [_c_vp],
    'hrnet_im2col_stem': [_c_int, _c_vp, _c_vp] + [_c_int] * 7 + [_c_vp],
    'hrnet_nhwc_to_nchw': [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp],
    'hrnet_nchw_to_nhwc': [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp],
    'hrnet_bias_grad': [_c_int, _c_vp, _c_vp, _c_vp] + [_c_int] * 4 + [_c_vp],
    'hrnet_fill_zero': [_c_vp, _c_i64, _c_vp],
    'hrnet_heatmap_loss_fwd': [_c_vp] * 4 + [_c_int] * 3 + [_c_vp],
    'hrnet_heatmap_loss_bwd': [_c_vp] * 4 + [_c_int] * 3 + [_c_vp],
    'hrnet_decode_expectation': [_c_vp, _c_vp] + [_c_int] * 3 + [_c_vp],
    'hrnet_decode_expectation_bwd': [_c_vp, _c_vp] + [_c_int] * 4 + [_c_vp],
    'hrnet_decode_argmax': [_c_vp] * 3 + [_c_int] * 4 + [_c_vp],
    'hrnet_joints_loss_fwd': [_c_vp] * 4 + [_c_int] * 2 + [_c_vp],
    'hrnet_joints_loss_bwd': [_c_vp] * 5 + [_c_int] * 2 + [_c_vp],
    'hrnet_adam_step': [_c_vp] * 4 + [_c_i64] + [_c_float] * 5 + [_c_int, _c_float, _c_vp],
    'hrnet_deform_conv_forward': [_c_vp] * 5 + [_c_int] * 15 + [_c_vp],
    'hrnet_deform_conv_wgrad_blocks': [_c_int] * 3,
    'hrnet_deform_conv_backward': [_c_vp] * 9 + [_c_int] * 15 + [_c_vp],
    'hrnet_modulated_deform_conv_forward': [_c_vp] * 6 + [_c_int] * 15 + [_c_vp],
    'hrnet_modulated_deform_conv_backward': [_c_vp] * 11 + [_c_int] * 15 + [_c_vp],
}
# plain-int helpers (no error code semantics)
_PLAIN = {'hrnet_abi_version', 'hrnet_ew_table_blocks', 'hrnet_conv_rows_bwdstats', 'hrnet_conv_route', 'hrnet_conv_ring_sum_enable', 'hrnet_conv_ring_enable', 'hrnet_conv_ring_supported', 'hrnet_conv_tiles', 'hrnet_conv_tile_walk', 'hrnet_conv_tiles_bwdstats', 'hrnet_wgrad_splits', 'hrnet_wgrad_tiles', 'hrnet_wgrad_blocks_per_split', 'hrnet_bwd_fused_supported', 'hrnet_bwd_fused_splits', 'hrnet_bwd_fused_kernel_name', 'hrnet_reduce_blocks',
          'hrnet_pack_blocks', 'hrnet_bwd_pw_supported', 'hrnet_bwd_pw_rows_supported', 'hrnet_bwd_pw_splits', 'hrnet_bwd_pw_kernel_name',
          'hrnet_conv_kernel_name', 'hrnet_wgrad_kernel_name', 'hrnet_conv_mode', 'hrnet_deform_conv_wgrad_blocks',
          'hrnet_head_mix_rows', 'hrnet_head_mix_supported'}
EXPORTED = sorted(list(_SIGS) + ['hrnet_last_error_string', 'hrnet_event_create'])

_lib = None


def lib():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libhrnet_hip.so not found at {} - build it with `python __graft_entry__.py` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback.'.format(LIB_PATH))
        # PyTorch ships its own libamdhip64; the library must bind to THAT copy (same soname), or the process ends up
        # with two HIP runtimes and this one's launches fail with "no ROCm-capable device is detected" (seen when
        # `python __graft_entry__.py smoke` loaded the library before anything had imported torch)
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        l.hrnet_last_error_string.argtypes = []
        l.hrnet_last_error_string.restype = ctypes.c_char_p
        l.hrnet_event_create.argtypes = []
        l.hrnet_event_create.restype = ctypes.c_void_p
        if l.hrnet_abi_version() != ABI_VERSION:
            raise RuntimeError('{} has ABI version {}, this host code needs {}: rebuild it with `python '
                               '__graft_entry__.py`'.format(LIB_PATH, l.hrnet_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def call(name, *args):
    """Call an entry point; negative return -> RuntimeError(hrnet_last_error_string())."""
    l = lib()
    rc = getattr(l, name)(*args)
    if name not in _PLAIN and rc != 0:
        raise RuntimeError('{} failed ({}): {}'.format(name, rc, l.hrnet_last_error_string().decode()))
    return rc


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def dtype_id(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return HR_F32
    if torch_dtype == torch.bfloat16:
        return HR_BF16
    raise ValueError('unsupported compute dtype {}'.format(torch_dtype))
