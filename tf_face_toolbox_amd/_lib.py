"""ctypes binding of libfte.so (include/fte.h).  This is the ONLY way arithmetic of the
training step runs: there is no CPU fallback.  If the library is missing or a call
fails, the error is raised -- never swallowed."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FTE_LIB') or os.path.join(_HERE, 'libfte.so')     # FTE_LIB: A/B builds of the same ABI


class FteError(RuntimeError):
    pass


_P = c_void_p
_SIGS = {
    'fte_version': (c_char_p, []),
    'fte_prof_enable': (c_int, [c_int]),
    'fte_prof_count': (c_int, []),
    'fte_prof_get': (c_int, [c_int, _P, _P, _P]),
    'fte_prof_get_shape': (c_int, [c_int, _P, _P]),
    'fte_prof_get_name': (c_int, [c_int, _P, c_int]),
    'fte_conv3x3_fwd': (c_int, [_P] * 7 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_conv3x3_fwd_ws_bytes': (c_size_t, [c_int] * 6),
    'fte_conv3x3_dgrad': (c_int, [_P] * 9 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_conv3x3_dgrad_ws_bytes': (c_size_t, [c_int] * 6),
    'fte_conv3x3_wgrad': (c_int, [_P] * 3 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_conv3x3_wgrad_ws_bytes': (c_size_t, [c_int] * 6),
    'fte_conv2d_fwd': (c_int, [_P] * 7 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_fwd_ws_bytes': (c_size_t, [c_int] * 7),
    'fte_conv2d_dgrad': (c_int, [_P] * 9 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_dgrad_ws_bytes': (c_size_t, [c_int] * 7),
    'fte_conv2d_wgrad': (c_int, [_P] * 3 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_wgrad_ws_bytes': (c_size_t, [c_int] * 7),
    'fte_im2col_first': (c_int, [_P] * 2 + [c_int] * 7 + [_P]),
    'fte_im2col_first_s16': (c_int, [_P] * 2 + [c_int] * 7 + [_P]),
    'fte_preprocess_u8': (c_int, [_P] * 2 + [c_int, c_long] + [c_int] * 5 + [_P]),
    'fte_bn_ws_bytes': (c_size_t, [c_int]),
    'fte_bn_train_fwd': (c_int, [_P] * 11 + [c_long, c_int, c_float, c_float, c_int, _P, c_size_t, _P]),
    'fte_bn_infer_fwd': (c_int, [_P] * 9 + [c_long, c_int, c_float, c_int, _P]),
    'fte_bn_train_bwd': (c_int, [_P] * 9 + [c_long, c_int, _P, c_size_t, _P]),
    'fte_bn_train_bwd_zmask': (c_int, [_P] * 10 + [c_long, c_int, _P, c_size_t, _P]),
    'fte_bn_train_bwd_res': (c_int, [_P] * 10 + [c_long, c_int, _P, c_size_t, _P]),
    'fte_bn_train_stats': (c_int, [_P] * 9 + [c_long, c_int, c_float, c_float, _P, c_size_t, _P]),
    'fte_bn_infer_coef': (c_int, [_P] * 6 + [c_int, c_float, _P]),
    'fte_relu_bwd': (c_int, [_P] * 3 + [c_long, _P]),
    'fte_maxpool3x3s2_fwd': (c_int, [_P] * 3 + [c_int] * 4 + [_P]),
    'fte_maxpool3x3s2_bwd': (c_int, [_P] * 3 + [c_int] * 4 + [_P]),
    'fte_gap_fwd': (c_int, [_P] * 2 + [c_int] * 3 + [_P]),
    'fte_gap_bwd': (c_int, [_P] * 2 + [c_int] * 3 + [_P]),
    'fte_dropout_fwd': (c_int, [_P] * 3 + [c_long, c_float, c_uint64, _P]),
    'fte_dropout_bwd': (c_int, [_P] * 3 + [c_long, c_float, _P]),
    'fte_gconv3x3_fwd': (c_int, [_P] * 3 + [c_int] * 6 + [_P]),
    'fte_gconv3x3_pack_bf16': (c_int, [_P] * 3 + [c_int, c_int, _P]),
    'fte_gconv3x3_bf16': (c_int, [_P] * 3 + [c_int] * 6 + [_P]),
    'fte_gconv3x3_wgrad_bf16_ws_bytes': (c_size_t, [c_int] * 6),
    'fte_gconv3x3_wgrad_bf16': (c_int, [_P] * 3 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_gconv3x3_dgrad': (c_int, [_P] * 3 + [c_int] * 6 + [_P]),
    'fte_gconv3x3_wgrad': (c_int, [_P] * 3 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_gconv3x3_wgrad_ws_bytes': (c_size_t, [c_int] * 6),
    'fte_bcast_add': (c_int, [_P] * 2 + [c_int] * 3 + [c_float, _P]),
    'fte_act_fwd': (c_int, [_P] * 2 + [c_long, c_int, _P]),
    'fte_act_bwd': (c_int, [_P] * 3 + [c_long, c_int, _P]),
    'fte_channel_scale_fwd': (c_int, [_P] * 3 + [c_int] * 3 + [_P]),
    'fte_channel_scale_bwd': (c_int, [_P] * 5 + [c_int] * 4 + [_P]),
    'fte_to_bf16': (c_int, [_P, _P, c_long, _P]),
    'fte_pack_weights_bf16': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'fte_pack_weights_bf16_table': (c_int, [_P, _P, _P, c_int, c_long, c_int, _P]),
    'fte_conv2d_fwd16': (c_int, [_P] * 8 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_dgrad16': (c_int, [_P] * 10 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_wgrad16': (c_int, [_P] * 3 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_fwd_s16': (c_int, [_P] * 9 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv2d_dgrad_s16': (c_int, [_P] * 9 + [c_int] * 7 + [_P, c_size_t, _P]),
    'fte_conv3x3_first_fwd_s16': (c_int, [_P] * 6 + [c_int] * 6 + [_P]),
    'fte_conv3x3_first_wgrad_s16': (c_int, [_P] * 3 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_bn_train_fwd_s16': (c_int, [_P] * 11 + [c_long, c_int, c_float, c_float, c_int, c_int, _P, c_size_t, _P]),
    'fte_bn_infer_fwd_s16': (c_int, [_P] * 9 + [c_long, c_int, c_float, c_int, c_int, _P]),
    'fte_bn_train_bwd_s16': (c_int, [_P] * 12 + [c_long, c_int, c_int, _P, c_size_t, _P]),
    'fte_relu_bwd_s16': (c_int, [_P] * 3 + [c_long, _P]),
    'fte_maxpool3x3s2_fwd_s16': (c_int, [_P] * 3 + [c_int] * 4 + [_P]),
    'fte_maxpool3x3s2_bwd_s16': (c_int, [_P] * 3 + [c_int] * 4 + [_P]),
    'fte_gap_fwd_s16': (c_int, [_P] * 2 + [c_int] * 3 + [_P]),
    'fte_gap_bwd_s16': (c_int, [_P] * 2 + [c_int] * 3 + [_P]),
    'fte_gconv3x3_bf16_s16': (c_int, [_P] * 3 + [c_int] * 6 + [_P]),
    'fte_gconv3x3_wgrad_bf16_s16': (c_int, [_P] * 3 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_channel_scale_fwd_s16': (c_int, [_P] * 3 + [c_int] * 3 + [_P]),
    'fte_channel_scale_bwd_s16': (c_int, [_P] * 4 + [c_int] * 4 + [_P]),
    'fte_channel_scale_bwd_apply_s16': (c_int, [_P] * 4 + [c_int] * 3 + [c_float, _P]),
    'fte_dense_small': (c_int, [_P] * 5 + [c_int] * 5 + [_P]),
    'fte_se_squeeze': (c_int, [_P] * 7 + [c_int] * 4 + [_P]),
    'fte_se_apply_fwd': (c_int, [_P] * 6 + [c_int] * 4 + [_P]),
    'fte_se_bwd_gate': (c_int, [_P] * 12 + [c_int] * 4 + [_P]),
    'fte_se_bn_bwd_coef': (c_int, [_P] * 11 + [c_int] * 3 + [_P]),
    'fte_se_bn_bwd_apply': (c_int, [_P] * 6 + [c_int] * 4 + [_P]),
    'fte_dwconv3x3_fwd_s16': (c_int, [_P] * 3 + [c_int] * 5 + [_P]),
    'fte_dwconv3x3_dgrad_s16': (c_int, [_P] * 3 + [c_int] * 5 + [_P]),
    'fte_dwconv3x3_wgrad_s16': (c_int, [_P] * 3 + [c_int] * 5 + [_P, c_size_t, _P]),
    'fte_channel_gather_s16': (c_int, [_P] * 4 + [c_long] + [c_int] * 3 + [_P]),
    'fte_channel_gather_affine_s16': (c_int, [_P] * 4 + [c_int, _P, _P, c_int, c_long, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P]),
    'fte_bn_train_stats_s16': (c_int, [_P] * 9 + [c_long, c_int, c_float, c_float, c_int, _P, c_size_t, _P]),
    'fte_conv2d_bn_fwd_ws_bytes': (c_size_t, [c_int] * 7),
    'fte_conv2d_bn_fwd': (c_int, [_P] * 11 + [c_float, c_float] + [_P] * 3 + [c_int] * 8 + [_P, c_size_t, _P]),
    'fte_conv2d_bn_fwd_folds': (c_int, [c_int] * 8),
    'fte_conv2d_dgrad_bn_ws_bytes': (c_size_t, [c_int] * 7),
    'fte_conv2d_dgrad_bn': (c_int, [_P] * 14 + [c_int] * 8 + [_P, c_size_t, _P]),
    'fte_bn_apply': (c_int, [_P] * 5 + [c_long, c_int, c_int, c_int, _P]),
    'fte_bn_bwd_apply': (c_int, [_P] * 4 + [c_long, c_int, c_int, _P]),
    'fte_gconv3x3_bn_ws_bytes': (c_size_t, [c_int] * 5),
    'fte_gconv3x3_bn_fwd_bf16_s16': (c_int, [_P] * 11 + [c_float, c_float] + [_P] * 3 + [c_int] * 5 + [_P, c_size_t, _P]),
    'fte_gconv3x3_dgrad_bn_bf16_s16': (c_int, [_P] * 12 + [c_int] * 5 + [_P, c_size_t, _P]),
    'fte_set_mfma_dtype': (c_int, [c_int]),
    'fte_get_mfma_dtype': (c_int, []),
    'fte_set_conv_algo': (c_int, [c_int]),
    'fte_conv3x3_algo': (c_int, [c_int] * 7),
    'fte_wino_pack_bytes': (c_size_t, [c_int] * 4),
    'fte_conv3x3_fwd_keep': (c_int, [_P] * 7 + [c_int] * 6 + [_P, _P, c_size_t, _P]),
    'fte_conv3x3_wgrad_kept': (c_int, [_P] * 3 + [c_int] * 6 + [_P, _P, c_size_t, _P]),
    'fte_get_conv_algo': (c_int, []),
    'fte_dwconv3x3_fwd': (c_int, [_P] * 3 + [c_int] * 5 + [_P]),
    'fte_dwconv3x3_dgrad': (c_int, [_P] * 3 + [c_int] * 5 + [_P]),
    'fte_dwconv3x3_wgrad': (c_int, [_P] * 3 + [c_int] * 5 + [_P, c_size_t, _P]),
    'fte_dwconv3x3_wgrad_ws_bytes': (c_size_t, [c_int] * 5),
    'fte_channel_gather': (c_int, [_P] * 4 + [c_long] + [c_int] * 3 + [_P]),
    'fte_channel_gather_affine': (c_int, [_P] * 4 + [c_int, _P, _P, c_int, c_long, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P]),
    'fte_conv3x3_first_fwd': (c_int, [_P] * 6 + [c_int] * 6 + [_P]),
    'fte_conv3x3_first_wgrad': (c_int, [_P] * 3 + [c_int] * 6 + [_P, c_size_t, _P]),
    'fte_conv3x3_first_wgrad_ws_bytes': (c_size_t, [c_int] * 6),
    'fte_gemm_nn': (c_int, [_P] * 4 + [c_int] * 3 + [_P, c_size_t, _P]),
    'fte_gemm_nn_act': (c_int, [_P] * 4 + [c_int] * 4 + [_P, c_size_t, _P]),
    'fte_gemm_nt': (c_int, [_P] * 4 + [c_int] + [_P] * 3 + [c_int] * 3 + [_P, c_size_t, _P]),
    'fte_gemm_tn': (c_int, [_P] * 3 + [c_int] * 3 + [_P, c_size_t, _P]),
    'fte_gemm_ws_bytes': (c_size_t, [c_int] * 3),
    'fte_softmax_ce_fwd_bwd': (c_int, [_P] * 4 + [c_int] * 3 + [c_float, _P]),
    'fte_focal_loss_fwd_bwd': (c_int, [_P] * 4 + [c_int] * 3 + [c_float] * 3 + [_P]),
    'fte_asoftmax_fwd_bwd': (c_int, [_P] * 4 + [c_float] + [_P] * 4 + [c_int] * 3 + [c_float, _P]),
    'fte_asoftmax_colcoef': (c_int, [_P] * 4 + [c_int] * 3 + [_P]),
    'fte_row_norms': (c_int, [_P] * 2 + [c_int] * 3 + [_P]),
    'fte_col_norms': (c_int, [_P] * 2 + [c_int] * 3 + [_P]),
    'fte_add_scaled_rows_cols': (c_int, [_P] * 4 + [c_int] * 3 + [_P]),
    'fte_flip_width': (c_int, [_P] * 2 + [c_int] * 4 + [_P]),
    'fte_axpby': (c_int, [c_float, _P, c_float, _P, _P, c_long, _P]),
    'fte_center_loss_fwd_bwd_update': (c_int, [_P] * 5 + [c_int] * 3 + [c_float] * 2 + [_P, c_size_t, _P]),
    'fte_center_scatter_update': (c_int, [_P] * 3 + [c_int] * 3 + [c_float, _P]),
    'fte_batch_hard_triplet_fwd_bwd': (c_int, [_P] * 2 + [c_float, c_int, c_float] + [_P] * 2 + [c_int] * 2 + [_P, c_size_t, _P]),
    'fte_reduce_rows': (c_int, [_P] * 3 + [c_int, c_long, c_long, c_int, c_float, _P]),
    'fte_sumsq': (c_int, [_P, c_long, c_float, _P, _P, c_size_t, _P]),
    'fte_sum': (c_int, [_P, c_long, c_float, _P, _P, c_size_t, _P]),
    'fte_momentum_update': (c_int, [_P] * 3 + [c_long] + [c_float] * 4 + [_P]),
    'fte_adam_update': (c_int, [_P] * 4 + [c_long] + [c_float] * 6 + [c_int, _P]),
}

_lib = None


def exported_names():
    return sorted(_SIGS)


_CODES = {-1: 'FTE_EINVAL -- a null pointer, a shape the kernel family does not tile (channel multiples, stride, '
              'ksize) or a tensor of 2 GiB or more (buffer-load range)',
          -2: 'FTE_EWORKSPACE -- workspace missing or smaller than the matching *_ws_bytes() query'}

MFMA_DTYPES = {'f32': 0, 'fp32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1, 'bf16s': 1}
# 'bf16s' = bf16 MFMA operands AND bf16 storage of the activations / inter-layer gradients (fte.h, "bf16 STORAGE"): the library's
# MFMA dtype is bf16; which entry points a net calls (the *_s16 ones) is the net's business and follows this flag
_storage16 = False


def load():
    """Load libfte.so; raises FteError with the build command when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FteError('libfte.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       '(or tf_face_toolbox_amd/csrc/build.sh). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing: loud on purpose
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    env = os.environ.get('FTE_MFMA_DTYPE')          # process-wide default of fte_set_mfma_dtype: f32 | bf16
    if env:
        if env not in MFMA_DTYPES:
            raise FteError('FTE_MFMA_DTYPE=%r: expected f32, bf16 or bf16s' % env)
        lib.fte_set_mfma_dtype(MFMA_DTYPES[env])
        global _storage16
        _storage16 = env == 'bf16s'
    return lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


_FN = {}
_Tensor = None


def call(name, *args):
    """Call an int-returning entry point; tensors are passed as device pointers.  (Kept lean: the graph nets make ~700 calls per
    step and a ShuffleNet-v2 step at <= 128 images per GPU is bound by this host loop, not by the GPU.)"""
    global _Tensor
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
        if _Tensor is None:
            import torch
            _Tensor = torch.Tensor
    T = _Tensor
    r = fn(*[a.data_ptr() if type(a) is T else (a.data_ptr() if hasattr(a, 'data_ptr') else a) for a in args])
    if r != 0:
        what = _CODES.get(r, 'hipError_t %d (see hip_runtime_api.h)' % r if r > 0 else 'unknown')
        raise FteError('%s failed with code %d: %s' % (name, r, what))


def query(name, *args):
    return getattr(load(), name)(*args)


def version():
    return load().fte_version().decode()


def source_stamp():
    """the hash of csrc/*.hip, *.h the loaded library was built from (csrc/build.sh); '' for an unstamped build"""
    v = version()
    return v.split('src:', 1)[1] if 'src:' in v else ''


def set_mfma_dtype(name):
    """'f32' (default, the reference's arithmetic) or 'bf16' (bf16 operands, fp32 accumulate and storage): fte.h."""
    if name not in MFMA_DTYPES:
        raise ValueError('unknown MFMA dtype %r (f32 | bf16 | bf16s)' % (name,))
    global _storage16
    call('fte_set_mfma_dtype', MFMA_DTYPES[name])
    _storage16 = name == 'bf16s'


def get_mfma_dtype():
    """'f32' or 'bf16' -- the operand precision of the MFMA products (storage: see bf16_storage())."""
    return 'bf16' if query('fte_get_mfma_dtype') == 1 else 'f32'


def bf16_storage():
    """True in the 'bf16s' mode: activations and inter-layer gradients live in HBM as bf16 (nets that implement it: SphereNet)."""
    return _storage16 and query('fte_get_mfma_dtype') == 1


def precision_mode():
    """'f32' | 'bf16' | 'bf16s' -- what set_mfma_dtype() was last given."""
    return 'bf16s' if bf16_storage() else get_mfma_dtype()


def prof_records(shapes=False):
    """All launch records since fte_prof_enable(1): list of (sig tuple, flops, ms) -- with `shapes`, of
    (sig tuple, flops, ms, (rows, N, K), algorithmic bytes, kernel symbol).  Synchronise first."""
    lib = load()
    out = []
    sig = (ctypes.c_int * 5)()
    mnk = (ctypes.c_int * 3)()
    fl = ctypes.c_double()
    by = ctypes.c_double()
    ms = ctypes.c_float()
    for i in range(lib.fte_prof_count()):
        r = lib.fte_prof_get(i, ctypes.cast(sig, c_void_p), ctypes.cast(ctypes.pointer(fl), c_void_p),
                             ctypes.cast(ctypes.pointer(ms), c_void_p))
        if r != 0:
            raise FteError('fte_prof_get(%d) failed with code %d' % (i, r))
        if shapes:
            r = lib.fte_prof_get_shape(i, ctypes.cast(mnk, c_void_p), ctypes.cast(ctypes.pointer(by), c_void_p))
            if r != 0:
                raise FteError('fte_prof_get_shape(%d) failed with code %d' % (i, r))
            name = ctypes.create_string_buffer(96)
            r = lib.fte_prof_get_name(i, ctypes.cast(name, c_void_p), 96)
            if r != 0:
                raise FteError('fte_prof_get_name(%d) failed with code %d' % (i, r))
            out.append((tuple(sig), fl.value, ms.value, tuple(mnk), by.value, name.value.decode()))
        else:
            out.append((tuple(sig), fl.value, ms.value))
    return out
