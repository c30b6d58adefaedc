"""ctypes binding of include/vaenpvc.h (one dlopen, no pybind / torch extension)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, '..', 'csrc'))
# VAENPVC_LIB: developer override used by scripts/build_variant.sh (kernel experiments)
LIB_PATH = os.environ.get('VAENPVC_LIB') or os.path.join(CSRC, 'libvaenpvc_hip.so')
MAX_LAYERS = 8
ABI_VERSION = 3

MODE_INFER, MODE_TRAIN = 0, 1
IMPL_AUTO, IMPL_GENERIC = 0, 1
PREC_BF16X3, PREC_BF16X2, PREC_BF16 = 3, 2, 1
# bf16 terms per fp32 operand on the bf16 matrix cores: 'bf16x2' (default, alias 'auto'): 16 mantissa bits per operand;
# 'bf16x3': fp32-exact; 'bf16': plain bf16 operands (the bf16 mode)
PRECISIONS = {'auto': 2, 'bf16x2': 2, 'bf16x3': 3, 'bf16': 1}


class HipVaeError(RuntimeError):
    pass


class Arch(C.Structure):
    _fields_ = [
        ('H', C.c_int32), ('z_dim', C.c_int32), ('y_dim', C.c_int32),
        ('n_enc', C.c_int32),
        ('enc_kernel', C.c_int32 * MAX_LAYERS), ('enc_stride', C.c_int32 * MAX_LAYERS),
        ('enc_output', C.c_int32 * MAX_LAYERS),
        ('gen_h', C.c_int32), ('gen_c', C.c_int32),
        ('n_dec', C.c_int32),
        ('dec_kernel', C.c_int32 * MAX_LAYERS), ('dec_stride', C.c_int32 * MAX_LAYERS),
        ('dec_output', C.c_int32 * MAX_LAYERS),
    ]


class DiscArch(C.Structure):
    _fields_ = [
        ('H', C.c_int32), ('n_layers', C.c_int32),
        ('kernel', C.c_int32 * MAX_LAYERS), ('stride', C.c_int32 * MAX_LAYERS), ('output', C.c_int32 * MAX_LAYERS),
    ]


# name -> (restype, argtypes); kept in one table so tests can check every symbol of
# include/vaenpvc.h is exported.
_P, _I64, _I32, _F, _U64 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64
# vaenpvc_bucket_cb(user, bucket, offset_floats, count_floats, ready_stream)
BUCKET_CB = C.CFUNCTYPE(None, _P, _I32, _I64, _I64, _P)
SIGNATURES = {
    'vaenpvc_abi_version': (C.c_int, []),
    'vaenpvc_last_error': (C.c_char_p, []),
    'vaenpvc_ctx_create': (C.c_int, [C.POINTER(Arch), C.POINTER(_P)]),
    'vaenpvc_ctx_destroy': (None, [_P]),
    'vaenpvc_set_impl': (C.c_int, [_P, C.c_int]),
    'vaenpvc_param_count': (C.c_int, [_P]),
    'vaenpvc_param_floats': (_I64, [_P]),
    'vaenpvc_param_info': (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(_I64), C.POINTER(_I32),
                                     C.POINTER(_I64)]),
    'vaenpvc_workspace_bytes': (_I64, [_P, _I64, C.c_int]),
    'vaenpvc_ws_find': (C.c_int, [_P, _I64, C.c_int, C.c_char_p, C.POINTER(_I64), C.POINTER(_I64)]),
    'vaenpvc_encode_fwd': (C.c_int, [_P, _P, _P, _I64, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_decode_fwd': (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, C.c_size_t, _P]),
    'vaenpvc_train_fwd_bwd': (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_loss_fwd': (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, C.c_size_t, _P]),
    'vaenpvc_adam_step': (C.c_int, [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _P]),
    'vaenpvc_adam_step_dev': (C.c_int, [_P, _P, _P, _P, _I64, _P, _F, _F, _F, _F, _F, _P]),
    'vaenpvc_tanhize_fwd': (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P]),
    'vaenpvc_tanhize_bwd': (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P]),
    'vaenpvc_set_tuned_masks': (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    'vaenpvc_timer_select': (C.c_int, [_P, C.c_char_p]),
    'vaenpvc_timer_read': (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    'vaenpvc_unpack_records': (C.c_int, [_P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    'vaenpvc_gather_unpack_records': (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    'vaenpvc_set_precision': (C.c_int, [_P, C.c_int]),
    'vaenpvc_get_precision': (C.c_int, [_P]),
    'vaenpvc_train_fwd_bwd_seeded': (C.c_int, [_P, _P, _P, _P, _U64, _U64, _P, _I64, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_loss_fwd_seeded': (C.c_int, [_P, _P, _P, _P, _U64, _U64, _I64, _P, _P, C.c_size_t, _P]),
    'vaenpvc_philox_normal': (C.c_int, [_U64, _U64, _P, _I64, _P]),
    'vaenpvc_set_bucket_callback': (C.c_int, [_P, BUCKET_CB, _P]),
    'vaenpvc_validate_ids': (C.c_int, [_P, _P, _I64, _P, _P]),
    'vaenpvc_summary': (C.c_int, [_P, _I64, _P, _I32, _P, _P, _P]),
    'vaenpvc_philox_uniform': (C.c_int, [_U64, _U64, _P, _I64, _P]),
    'vaenpvc_train_fwd_bwd_target': (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_train_bwd_target': (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_disc_create': (C.c_int, [C.POINTER(DiscArch), C.POINTER(_P)]),
    'vaenpvc_disc_destroy': (None, [_P]),
    'vaenpvc_disc_param_count': (C.c_int, [_P]),
    'vaenpvc_disc_param_floats': (_I64, [_P]),
    'vaenpvc_disc_param_info': (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(_I64), C.POINTER(_I32),
                                          C.POINTER(_I64)]),
    'vaenpvc_disc_workspace_bytes': (_I64, [_P, _I64]),
    'vaenpvc_disc_fwd': (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_disc_critic_fwd_bwd': (C.c_int, [_P, _P, _P, _P, _P, _I64, _F, _P, _P, _P, C.c_size_t, _P]),
    'vaenpvc_disc_generator_target': (C.c_int, [_P, _P, _P, _P, _I64, _F, _P, _P, _P, C.c_size_t, _P]),
}

_lib = None


def build_library(verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (in-tree, so the .so travels with the repo)."""
    cmd = ['make', '-C', CSRC, '-j8']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise HipVaeError('hipcc build of libvaenpvc_hip.so failed')
    return LIB_PATH


def load_library():
    """dlopen libvaenpvc_hip.so and bind every entry point.  Fails loudly when the
    library is missing: there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipVaeError(
            'libvaenpvc_hip.so is not built (%s). Run `make -C vae-npvc_amd/csrc` or '
            '`python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.vaenpvc_abi_version()
    if v != ABI_VERSION:
        raise HipVaeError('ABI version mismatch: library %d, binding %d' % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load_library().vaenpvc_last_error()
        raise HipVaeError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))
