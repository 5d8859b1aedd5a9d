"""ctypes binding of libsatcv.so (the C-ABI HIP library, include/satcv.h).

The product path has NO fallback: if the shared library is missing or a symbol
declared in include/satcv.h is absent, import fails loudly.
"""
import ctypes as C
import os

# The library takes device pointers and streams from torch, so both must sit on ONE HIP runtime:
# import torch first (it loads its bundled libamdhip64) so that libsatcv.so binds to that copy.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SATCV_LIB') or os.path.join(_HERE, 'libsatcv.so')     # SATCV_LIB: profiling variants only

F32, BF16, FP8, FP8X, F64 = 0, 1, 2, 3, 4
STAT_ROWS = 32

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [('x0', c_vp), ('x1', c_vp), ('c0', c_i32), ('c1', c_i32),
                ('in_scale', c_vp), ('in_shift', c_vp), ('in_relu', c_i32),
                ('w', c_vp), ('bias', c_vp), ('y', c_vp), ('ldy', c_i32),
                ('stats', c_vp), ('stats_ld', c_i32),
                ('n', c_i32), ('h', c_i32), ('w_', c_i32),
                ('cout', c_i32), ('cout_pad', c_i32),
                ('kh', c_i32), ('kw', c_i32), ('dil', c_i32),
                ('mode_in', c_i32), ('mode_out', c_i32), ('f', c_i32),
                ('cstat', c_i32), ('out_relu', c_i32), ('dtype', c_i32), ('accumulate', c_i32),
                ('stride', c_i32), ('hin', c_i32), ('win', c_i32), ('out_scale', c_vp), ('pool_y', c_vp), ('pool_ld', c_i32), ('pool_f', c_i32),
                ('bst_y', c_vp), ('bst_y1', c_vp), ('bst_ld', c_i32), ('bst_ld1', c_i32), ('bst_split', c_i32),
                ('bst_scale', c_vp), ('bst_shift', c_vp), ('bst_mean', c_vp), ('bst_rstd', c_vp), ('bst_relu', c_i32), ('tile_policy', c_i32)]


class PackJob(C.Structure):
    _fields_ = [('src', c_vp), ('dst', c_vp), ('mode', c_i32), ('taps', c_i32), ('cin', c_i32), ('cout', c_i32), ('kpad', c_i32), ('npad', c_i32)]


class TileDesc(C.Structure):
    _fields_ = [('src', c_vp), ('src_kind', c_i32), ('n', c_i32), ('c', c_i32), ('hin', c_i32), ('win', c_i32), ('h', c_i32), ('w_', c_i32),
                ('rescale', C.c_double), ('nan_mask', c_i32), ('replace', c_i32), ('seed', C.c_uint64), ('ch_mean', c_vp),
                ('contra_mul', C.c_double), ('bright_mul', C.c_double), ('flip_v', c_i32), ('flip_h', c_i32), ('rot', c_i32),
                ('dst', c_vp), ('ldc', c_i32), ('coff', c_i32)]


class WgradDesc(C.Structure):
    _fields_ = [('x0', c_vp), ('x1', c_vp), ('c0', c_i32), ('c1', c_i32),
                ('in_scale', c_vp), ('in_shift', c_vp), ('in_relu', c_i32),
                ('dy', c_vp), ('lddy', c_i32), ('dw', c_vp),
                ('cin', c_i32), ('cout', c_i32),
                ('n', c_i32), ('h', c_i32), ('w_', c_i32),
                ('kh', c_i32), ('kw', c_i32), ('dil', c_i32),
                ('mode_dy', c_i32), ('f', c_i32), ('transposed', c_i32),
                ('workspace', c_vp), ('workspace_bytes', c_i64), ('dtype', c_i32), ('accumulate', c_i32), ('whole_chip', c_i32), ('defer_reduce', c_i32)]


class BwdfDesc(C.Structure):
    _fields_ = [('g', c_vp), ('yraw', c_vp), ('ldg', c_i32),
                ('bn_scale', c_vp), ('bn_shift', c_vp), ('bn_mean', c_vp), ('bn_rstd', c_vp), ('bn_coef', c_vp), ('linear', c_i32),
                ('x0', c_vp), ('x1', c_vp), ('c0', c_i32), ('c1', c_i32),
                ('in_scale', c_vp), ('in_shift', c_vp), ('in_relu', c_i32),
                ('w_dgrad', c_vp), ('dx', c_vp), ('lddx', c_i32),
                ('dw', c_vp), ('cin', c_i32), ('cout', c_i32),
                ('n', c_i32), ('h', c_i32), ('w_', c_i32), ('kh', c_i32), ('kw', c_i32), ('dil', c_i32),
                ('workspace', c_vp), ('workspace_bytes', c_i64), ('dtype', c_i32), ('accumulate', c_i32),
                ('bst_sums', c_vp), ('bst_sums_ld', c_i32), ('bst_mean', c_vp), ('bst_rstd', c_vp), ('bst_act_form', c_i32),
                ('dpool', c_vp), ('lddp', c_i32), ('amax', c_vp),
                ('hg_dlogits', c_vp), ('hg_w', c_vp), ('hg_ncls', c_i32), ('defer_reduce', c_i32)]


class CtbfDesc(C.Structure):
    _fields_ = [('g', c_vp), ('ldg', c_i32), ('yup', c_vp), ('ldy', c_i32),
                ('bn_scale', c_vp), ('bn_shift', c_vp), ('bn_mean', c_vp), ('bn_rstd', c_vp), ('bn_c1', c_vp), ('bn_c2', c_vp), ('linear', c_i32),
                ('x', c_vp), ('ldx', c_i32), ('in_scale', c_vp), ('in_shift', c_vp), ('in_relu', c_i32),
                ('w_dgrad', c_vp), ('w_npad', c_i32), ('dx', c_vp), ('lddx', c_i32), ('dw', c_vp), ('cin', c_i32), ('cout', c_i32),
                ('n', c_i32), ('h', c_i32), ('w_', c_i32), ('f', c_i32), ('workspace', c_vp), ('workspace_bytes', c_i64),
                ('dtype', c_i32), ('accumulate', c_i32), ('defer_reduce', c_i32),
                ('bst_sums', c_vp), ('bst_sums_ld', c_i32), ('bst_mean', c_vp), ('bst_rstd', c_vp)]


class ReduceJob(C.Structure):
    _fields_ = [('ws', c_vp), ('dw', c_vp), ('nslab', c_i32), ('taps', c_i32), ('kpad', c_i32), ('npad', c_i32), ('cin', c_i32), ('nvalid', c_i32),
                ('transposed', c_i32), ('accumulate', c_i32), ('lanes', c_i32), ('pad_', c_i32)]


class BnBwdDesc(C.Structure):
    _fields_ = [('da', c_vp), ('ldda', c_i32), ('dpool', c_vp), ('lddp', c_i32), ('f', c_i32),
                ('yraw', c_vp), ('ldy', c_i32),
                ('scale', c_vp), ('shift', c_vp), ('mean', c_vp), ('rstd', c_vp),
                ('sums', c_vp), ('sums_ld', c_i32), ('coef', c_vp),
                ('dy', c_vp), ('lddy_out', c_i32), ('dbias', c_vp),
                ('n', c_i32), ('h', c_i32), ('w_', c_i32), ('c', c_i32), ('dtype', c_i32), ('linear', c_i32),
                ('yraw1', c_vp), ('ldy1', c_i32), ('dy1', c_vp), ('lddy1', c_i32), ('c_split', c_i32),
                ('sk_sums', c_vp), ('sk_sums_ld', c_i32)]


class HeadDesc(C.Structure):
    _fields_ = [('x', c_vp), ('ldx', c_i32), ('cin', c_i32),
                ('in_scale', c_vp), ('in_shift', c_vp), ('w', c_vp), ('b', c_vp),
                ('ncls', c_i32), ('activation', c_i32), ('thresh', c_f32),
                ('probs', c_vp), ('classes', c_vp), ('dlogits', c_vp),
                ('dx', c_vp), ('lddx', c_i32), ('dw', c_vp), ('db', c_vp),
                ('bnr_mean', c_vp), ('bnr_rstd', c_vp), ('bnr_sums', c_vp), ('bnr_sums_ld', c_i32),
                ('npix', c_i64), ('dtype', c_i32), ('partials', c_vp)]


class LstmGatesDesc(C.Structure):
    _fields_ = [('xg', c_vp), ('ldx', c_i32), ('hg', c_vp), ('ldh_g', c_i32), ('c_prev', c_vp),
                ('c_out', c_vp), ('h_out', c_vp), ('ldh', c_i32), ('gates_out', c_vp),
                ('stats', c_vp), ('stats_ld', c_i32),
                ('dh_a', c_vp), ('lddh_a', c_i32), ('dh_b', c_vp), ('lddh_b', c_i32), ('dc_next', c_vp),
                ('dz_out', c_vp), ('lddz', c_i32), ('dc_prev_out', c_vp),
                ('npix', c_i64), ('filters', c_i32), ('rec_act', c_i32), ('act', c_i32), ('dtype', c_i32)]


class DenseSrc(C.Structure):
    _fields_ = [('x', c_vp), ('ld', c_i32), ('cin', c_i32), ('dtype', c_i32),
                ('in_scale', c_vp), ('in_shift', c_vp), ('in_relu', c_i32),
                ('hs', c_i32), ('ws', c_i32),
                ('dx', c_vp), ('lddx', c_i32), ('dx_dtype', c_i32)]


class DenseDesc(C.Structure):
    _fields_ = [('src', DenseSrc * 2), ('nsrc', c_i32),
                ('w', c_vp), ('b', c_vp), ('cout', c_i32),
                ('activation', c_i32), ('max_value', c_f32),
                ('out', c_vp), ('classes', c_vp), ('z_out', c_vp),
                ('npix', c_i64), ('h', c_i32), ('w_', c_i32),
                ('dout', c_vp), ('dz_out', c_vp), ('dw', c_vp), ('db', c_vp)]


# name -> (restype, argtypes); every symbol declared in include/satcv.h
_SIGS = {
    'satcv_version': (C.c_char_p, []),
    'satcv_last_error': (C.c_char_p, []),
    'satcv_device_info': (C.c_int, [C.POINTER(c_i32)]),
    'satcv_ingest_nhwc': (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    'satcv_ingest_nhwc_scaled': (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_vp]),
    'satcv_ingest_chw': (C.c_int, [c_vp, c_i32, c_f32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_pack_weights': (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_pack_job_items': (c_i64, [C.POINTER(PackJob)]),
    'satcv_pack_weights_batched': (C.c_int, [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    'satcv_conv2d_igemm': (C.c_int, [C.POINTER(ConvDesc), c_vp]),
    'satcv_conv2d_igemm_pipelined': (C.c_int, [C.POINTER(ConvDesc)]),
    'satcv_conv2d_wgrad_workspace': (c_i64, [C.POINTER(WgradDesc)]),
    'satcv_conv2d_wgrad': (C.c_int, [C.POINTER(WgradDesc), c_vp]),
    'satcv_conv2d_wgrad_reduce_job': (C.c_int, [C.POINTER(WgradDesc), C.POINTER(ReduceJob)]),
    'satcv_reduce_job_items': (c_i64, [C.POINTER(ReduceJob)]),
    'satcv_reduce_slabs_batched': (C.c_int, [c_vp, c_vp, c_i32, c_i64, c_vp]),
    'satcv_conv2d_bwd_fused_workspace': (c_i64, [C.POINTER(BwdfDesc)]),
    'satcv_conv2d_bwd_fused': (C.c_int, [C.POINTER(BwdfDesc), c_vp]),
    'satcv_conv2d_bwd_fused_reduce_job': (C.c_int, [C.POINTER(BwdfDesc), C.POINTER(ReduceJob)]),
    'satcv_convt_bwd_fused_workspace': (c_i64, [C.POINTER(CtbfDesc)]),
    'satcv_convt_bwd_fused': (C.c_int, [C.POINTER(CtbfDesc), c_vp]),
    'satcv_convt_bwd_fused_reduce_job': (C.c_int, [C.POINTER(CtbfDesc), C.POINTER(ReduceJob)]),
    'satcv_bn_finalize_train': (C.c_int, [c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_f32, c_f32, c_i32, c_i32,
                                          c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'satcv_bn_affine_infer': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp]),
    'satcv_bn_affine_infer_batched': (C.c_int, [c_vp, c_i32, c_f32, c_vp]),
    'satcv_bn_relu_pool': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_bn_relu_pool_amax': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_bn_bwd_reduce': (C.c_int, [C.POINTER(BnBwdDesc), c_vp]),
    'satcv_bn_bwd_finalize': (C.c_int, [c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'satcv_bn_bwd_apply': (C.c_int, [C.POINTER(BnBwdDesc), c_vp]),
    'satcv_bn_bwd_finalize2': (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'satcv_maxpool': (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_affine_requant': (C.c_int, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp]),
    'satcv_add_act': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp]),
    'satcv_relu_bwd': (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    'satcv_bias_grad_workspace': (c_i64, [c_i64, c_i32]),
    'satcv_bias_grad': (C.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'satcv_upsample_head': (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    'satcv_dropout_mask': (C.c_int, [C.c_uint64, C.c_uint64, c_f32, c_i64, c_vp, c_vp]),
    'satcv_dropout_apply': (C.c_int, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_tile_channel_mean': (C.c_int, [C.POINTER(TileDesc), c_vp, c_vp]),
    'satcv_tile_ingest': (C.c_int, [C.POINTER(TileDesc), c_vp]),
    'satcv_label_onehot': (C.c_int, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp]),
    'satcv_crc32c': (C.c_uint32, [c_vp, C.c_uint64, C.c_uint32]),
    'satcv_head_fwd': (C.c_int, [C.POINTER(HeadDesc), c_vp]),
    'satcv_head_bwd': (C.c_int, [C.POINTER(HeadDesc), c_vp]),
    'satcv_head_bwd_workspace': (c_i64, [C.POINTER(HeadDesc)]),
    'satcv_head_bwd_finalize': (C.c_int, [C.POINTER(HeadDesc), c_vp]),
    'satcv_loss_fwd_bwd': (C.c_int, [c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_f32, c_vp, c_vp, c_vp]),
    'satcv_loss_global_fwd_bwd': (C.c_int, [c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    'satcv_confusion': (C.c_int, [c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    'satcv_adam_step': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp]),
    'satcv_adam_step_part': (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp, c_vp, c_i32, c_vp]),
    'satcv_comm_unique_id': (C.c_int, [c_vp]),
    'satcv_comm_init': (C.c_int, [C.POINTER(c_vp), c_i32, c_i32, c_vp]),
    'satcv_comm_destroy': (C.c_int, [c_vp]),
    'satcv_comm_info': (C.c_int, [c_vp, C.POINTER(c_i32), C.POINTER(c_i32)]),
    'satcv_allreduce_grads': (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp]),
    'satcv_allreduce': (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    'satcv_ingest_seq': (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'satcv_convlstm_gates_fwd': (C.c_int, [C.POINTER(LstmGatesDesc), c_vp]),
    'satcv_convlstm_gates_bwd': (C.c_int, [C.POINTER(LstmGatesDesc), c_vp]),
    'satcv_dense_small_fwd': (C.c_int, [C.POINTER(DenseDesc), c_vp]),
    'satcv_dense_small_bwd': (C.c_int, [C.POINTER(DenseDesc), c_vp]),
    'satcv_zero2': (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    'satcv_graph_begin': (C.c_int, [c_vp]),
    'satcv_graph_end': (C.c_int, [c_vp, C.POINTER(c_vp)]),
    'satcv_graph_launch': (C.c_int, [c_vp, c_vp]),
    'satcv_graph_destroy': (C.c_int, [c_vp]),
    'satcv_prof_enable': (C.c_int, [c_i32]),
    'satcv_prof_collect': (C.c_int, [c_i32, C.POINTER(C.c_double), C.POINTER(c_i64), C.POINTER(C.c_double)]),
    'satcv_set_option': (C.c_int, [C.c_char_p, c_i32]),
    'satcv_get_option': (C.c_int, [C.c_char_p, C.POINTER(c_i32)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


class SatcvError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
            'Build it with `python -m satellite_computervision_amd.build` or __graft_entry__.build().')
    hip = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise SatcvError(f'satcv error {rc}: {lib.satcv_last_error().decode()}')


def call(name, *args):
    check(getattr(lib, name)(*args))
