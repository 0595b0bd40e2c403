"""ctypes binding of libiodine_hip.so (the C ABI declared in include/iodine_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails the
error is raised, never papered over with a PyTorch / CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IODINE_HIP_LIB selects another build of the same library (tools/ab_libs.sh: same-box A/B of two builds)
LIB_PATH = os.environ.get('IODINE_HIP_LIB') or os.path.join(_HERE, 'libiodine_hip.so')

ENC_ORDER = ('posterior', 'grad_post', 'image', 'means', 'mask', 'mask_logits', 'mask_posterior',
             'grad_means', 'grad_mask', 'likelihood', 'leave_one_out_likelihood', 'coordinate')
ENC_FULL = 0xFFF

# every symbol include/iodine_hip.h declares
EXPORTS = (
    'iodine_abi_version', 'iodine_create', 'iodine_destroy', 'iodine_last_error', 'iodine_num_params',
    'iodine_param_info', 'iodine_set_params', 'iodine_workspace_bytes', 'iodine_set_workspace',
    'iodine_reconstruct', 'iodine_decode', 'iodine_elbo', 'iodine_last_elbo_outputs', 'iodine_last_posterior', 'iodine_randn',
    'iodine_train_forward', 'iodine_train_backward', 'iodine_train_backward_flat', 'iodine_logger_scalars',
    'iodine_adam_step', 'iodine_ari_table', 'iodine_set_option', 'iodine_profile_read', 'iodine_debug_copy', 'iodine_linspace_host', 'iodine_op_conv3x3', 'iodine_op_dec_out',
    'iodine_op_conv3x3_wgrad', 'iodine_op_conv3x3_wgrad_f32', 'iodine_op_dec_out_f16x3', 'iodine_op_gen_conv',
)


class Config(C.Structure):
    _fields_ = [
        ('dim_latent', C.c_int), ('iters', C.c_int), ('slots', C.c_int), ('img_size', C.c_int),
        ('img_channels', C.c_int), ('sigma', C.c_double), ('layernorm', C.c_int), ('stop_gradient', C.c_int),
        ('encoding', C.c_uint), ('ref_conv_chan', C.c_int), ('ref_conv_layers', C.c_int),
        ('ref_mlp_units', C.c_int), ('ref_kernel_size', C.c_int), ('ref_stride', C.c_int),
        ('dec_conv_chan', C.c_int), ('dec_conv_layers', C.c_int), ('dec_kernel_size', C.c_int),
    ]


_lib = None
_FP = C.POINTER(C.c_float)


def encoding_bits(names) -> int:
    bits = 0
    for n in names:
        if n not in ENC_ORDER:
            raise ValueError(f'unknown ARCH.ENCODING entry {n!r}')
        bits |= 1 << ENC_ORDER.index(n)
    return bits


def lib() -> C.CDLL:
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -m iodine_amd.build` '
                           '(there is no CPU / PyTorch fallback for this path)')
    # torch must be imported first: its bundled libamdhip64 (same soname) then serves both torch and this
    # library, so streams and device pointers are shared within ONE HIP runtime instance.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.iodine_abi_version.restype = ci
    L.iodine_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.iodine_destroy.argtypes = [vp]
    L.iodine_destroy.restype = None
    L.iodine_last_error.argtypes = [vp]
    L.iodine_last_error.restype = C.c_char_p
    L.iodine_num_params.argtypes = [vp]
    L.iodine_param_info.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(ci), C.POINTER(C.c_longlong)]
    L.iodine_set_params.argtypes = [vp, vp, C.POINTER(vp), ci]
    L.iodine_workspace_bytes.argtypes = [vp, ci, ci]
    L.iodine_workspace_bytes.restype = C.c_size_t
    L.iodine_set_workspace.argtypes = [vp, vp, C.c_size_t]
    L.iodine_reconstruct.argtypes = [vp, vp, ci] + [vp] * 9
    L.iodine_decode.argtypes = [vp, vp, ci] + [vp] * 4
    L.iodine_elbo.argtypes = [vp, vp, ci] + [vp] * 5
    L.iodine_last_elbo_outputs.argtypes = [vp, vp, ci] + [vp] * 5
    L.iodine_last_posterior.argtypes = [vp, vp, ci, vp, vp]
    L.iodine_randn.argtypes = [vp, vp, C.c_longlong, C.c_ulonglong, C.c_ulonglong]
    L.iodine_train_forward.argtypes = [vp, vp, ci] + [vp] * 4
    L.iodine_train_backward.argtypes = [vp, vp, cf, C.POINTER(vp), ci]
    L.iodine_train_backward_flat.argtypes = [vp, vp, vp, vp, ci]
    L.iodine_logger_scalars.argtypes = [vp, vp, vp]
    L.iodine_adam_step.argtypes = [vp, vp, vp, ci, C.c_longlong] + [C.c_double] * 5 + [ci]
    L.iodine_ari_table.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
    L.iodine_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.iodine_profile_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong), ci]
    L.iodine_debug_copy.argtypes = [vp, vp, C.c_char_p, ci, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.iodine_linspace_host.argtypes = [ci, _FP]
    L.iodine_linspace_host.restype = None
    L.iodine_op_conv3x3.argtypes = [vp, ci] + [vp] * 5 + [ci] * 10
    L.iodine_op_dec_out.argtypes = [vp] + [vp] * 4 + [ci] * 3
    L.iodine_op_conv3x3_wgrad.argtypes = [vp] + [vp] * 4 + [ci] * 6
    if hasattr(L, 'iodine_op_dec_out_f16x3'):
        L.iodine_op_dec_out_f16x3.argtypes = [vp] + [vp] * 4 + [ci] * 4
    if hasattr(L, 'iodine_op_gen_conv'):
        L.iodine_op_gen_conv.argtypes = [vp, ci] + [vp] * 6 + [ci] * 8
    if hasattr(L, 'iodine_op_conv3x3_wgrad_f32'):       # (absent from older builds loaded through IODINE_HIP_LIB for same-box A/B)
        L.iodine_op_conv3x3_wgrad_f32.argtypes = [vp] + [vp] * 4 + [ci] * 3
    if L.iodine_abi_version() != 3:
        raise RuntimeError('libiodine_hip.so ABI version mismatch')
    _lib = L
    return L


def check(rc: int, handle=None, what: str = ''):
    if rc != 0:
        msg = lib().iodine_last_error(handle)
        raise RuntimeError(f'{what or "libiodine_hip"} failed (status {rc}): '
                           f'{msg.decode() if msg else "unknown error"}')


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return C.c_void_p(t.data_ptr())
