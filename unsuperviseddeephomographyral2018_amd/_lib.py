"""ctypes binding of include/uh_hotpath.h.  There is NO fallback: if the library is missing or a
symbol is absent, importing/using the ops raises -- the product path never computes on CPU."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# UH_LIB_PATH lets a developer A/B a differently-built copy of the SAME library (tools/); it is not a fallback.
LIB_PATH = os.environ.get('UH_LIB_PATH') or os.path.join(HERE, 'lib', 'libuh_hotpath.so')

UH_ABI_VERSION = 8
UH_DLT_SOLVE_F32 = 0
UH_DLT_SOLVE_F64 = 1
UH_DLT_ZERO_NONFINITE_GRAD = 8
UH_TAIL_FUSED_PATCH = 2
UH_TAIL_GRAPH = 4
LOSS_KINDS = {'rec_loss': 0, 'ssim_loss': 1, 'l1_loss': 2, 'l1_smooth_loss': 3, 'ncc_loss': 4}
KERNEL_COUNT = 17

_p = C.c_void_p
_i = C.c_int
_u = C.c_uint
_z = C.c_size_t

# name -> (restype, argtypes); mirrors include/uh_hotpath.h one to one
SIGNATURES = {
    'uh_abi_version': (_i, []),
    'uh_error_string': (C.c_char_p, [_i]),
    'uh_dlt_forward': (_i, [_p, _p, _p, _p, _p, _p, _i, _u, _p]),
    'uh_dlt_backward': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _u, _p]),
    'uh_dlt_zeroed_pairs': (_i, [_p, _i, _p]),
    'uh_warp_forward': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'uh_warp_forward_literal': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'uh_warp_backward_workspace_bytes': (_z, [_i, _i, _i, _i, _i, _i]),
    'uh_warp_backward': (_i, [_p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _p]),
    'uh_warp_patch_backward_workspace_bytes': (_z, [_i, _i, _i, _i]),
    'uh_warp_patch_backward': (_i, [_p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _p]),
    'uh_gather_patch_losses_forward': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _p]),
    'uh_warp_patch_loss_backward': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _p]),
    'uh_gray_patch_forward': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'uh_gray_patch_backward': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'uh_l1_loss_workspace_bytes': (_z, [_z]),
    'uh_l1_loss_forward': (_i, [_p, _p, _p, _p, _z, _z, _p]),
    'uh_l1_loss_backward': (_i, [_p, _p, _p, _p, _z, _p]),
    'uh_prepare_inputs': (_i, [_p] * 13 + [_i, _i, _i, _i, _p]),
    'uh_patch_losses_workspace_bytes': (_z, [_i, _i]),
    'uh_patch_losses_forward': (_i, [_p, _p, _p, _p, _p, _p, _z, _i, _i, _p]),
    'uh_patch_loss_backward': (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _p]),
    'uh_warp_patch_l1_workspace_bytes': (_z, [_i, _i]),
    'uh_warp_patch_l1_fwdbwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _p]),
    'uh_tail_create': (_i, [_p, _i, _i, _i, _i, _i, _u]),
    'uh_tail_workspace_bytes': (_z, [_p]),
    'uh_tail_warped_offset': (_z, [_p]),
    'uh_tail_run': (_i, [_p] * 12 + [_p, _z, _p]),
    'uh_tail_stats': (_i, [_p, _p, _p]),
    'uh_tail_destroy': (None, [_p]),
    'uh_relu_mask_bytes': (_z, [_z, _i]),
    'uh_bias_relu_forward': (_i, [_p, _p, _p, _z, _i, _p]),
    'uh_bias_relu_backward_workspace_bytes': (_z, [_z, _i]),
    'uh_bias_relu_backward': (_i, [_p, _p, _p, _p, _p, _z, _z, _i, _p]),
    'uh_pool_mask_bytes': (_z, [_i, _i, _i, _i]),
    'uh_bias_relu_pool_forward': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'uh_bias_relu_pool_backward_workspace_bytes': (_z, [_i, _i, _i, _i]),
    'uh_bias_relu_pool_backward': (_i, [_p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _p]),
    'uh_profile_enable': (_i, [_i]),
    'uh_profile_read': (_i, [_p, _p]),
    'uh_kernel_name': (C.c_char_p, [_i]),
}

_lib = None


class UHError(RuntimeError):
    pass


def load():
    """Load libuh_hotpath.so (once) and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UHError('%s not found -- run `python -c "import __graft_entry__ as g; g.build()"` '
                      '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.uh_abi_version() != UH_ABI_VERSION:
        raise UHError('ABI version mismatch: library %d, binding %d' % (lib.uh_abi_version(), UH_ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise UHError('%s failed: %s (code %d)' % (what, load().uh_error_string(code).decode(), code))


KERNEL_IDS = {'dlt_forward': 0, 'dlt_backward': 1, 'warp_forward': 2, 'warp_backward': 3, 'warp_backward_finish': 4,
              'gray_patch_forward': 5, 'gray_patch_backward': 6, 'l1_forward': 7, 'l1_backward': 8,
              'warp_patch_l1_fused': 9, 'warp_patch_l1_finish': 10, 'patch_losses': 11, 'patch_losses_finish': 12,
              'prepare_inputs': 13, 'bias_relu_forward': 14, 'bias_relu_backward': 15,
              'patch_loss_backward': 16}


def dlt_zeroed_pairs(reset=False, stream=None):
    """Pairs whose d loss / d pred_h4p UH_DLT_ZERO_NONFINITE_GRAD zeroed on the current device since the last reset.
    Synchronous on `stream` (a raw hipStream_t value; None = torch's current stream) -- log time only, and never while that
    stream is being captured (the library refuses with UH_E_CAPTURING instead of invalidating the capture)."""
    if stream is None:
        import torch
        stream = torch.cuda.current_stream().cuda_stream
    n = C.c_ulonglong(0)
    check(load().uh_dlt_zeroed_pairs(C.byref(n), 1 if reset else 0, C.c_void_p(stream)), 'uh_dlt_zeroed_pairs')
    return int(n.value)


def profile_enable(on, only=None):
    """on: bool.  only: iterable of kernel names (KERNEL_IDS) to time; None = every kernel."""
    if not on:
        return load().uh_profile_enable(0)
    if only is None:
        return load().uh_profile_enable(1)
    mask = 0
    for name in only:
        mask |= 1 << (KERNEL_IDS[name] + 1)
    return load().uh_profile_enable(mask)


def profile_read():
    """-> {kernel_name: (total_ms, launches)}"""
    lib = load()
    ms = (C.c_double * KERNEL_COUNT)()
    n = (C.c_longlong * KERNEL_COUNT)()
    check(lib.uh_profile_read(ms, n), 'uh_profile_read')
    return {lib.uh_kernel_name(k).decode(): (ms[k], n[k]) for k in range(KERNEL_COUNT)}
