"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/uh_hotpath.h
declares, validates arguments without touching a GPU, and the product path refuses CPU tensors."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'uh_hotpath.h')).read()
    return sorted(set(re.findall(r'^UH_API\s+[\w\s\*]+?\b(uh_\w+)\s*\(', src, flags=re.M)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ('uh_dlt_forward', 'uh_dlt_backward', 'uh_warp_forward', 'uh_warp_backward',
                 'uh_warp_backward_workspace_bytes', 'uh_warp_patch_l1_fwdbwd', 'uh_patch_losses_forward',
                 'uh_patch_losses_workspace_bytes', 'uh_gray_patch_forward',
                 'uh_l1_loss_forward'):
        assert must in syms
    assert len(syms) >= 17


def test_header_is_plain_c_and_a_c_caller_links(uh_lib_path, tmp_path):
    """include/uh_hotpath.h is the drop-in boundary for a C caller (cgo / JNI / ctypes ...): it must compile as C99 with
    gcc -- no C++, no HIP, no torch types -- and a C program that takes the address of every declared entry point must link
    against the shared library."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    syms = declared_symbols()
    src = tmp_path / 'caller.c'
    src.write_text('#include "uh_hotpath.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\nint main(void) {\n'
                   '  fn_t f[] = {%s};\n  printf("%%d %%d\\n", (int)(sizeof f / sizeof f[0]), UH_ABI_VERSION);\n'
                   '  return f[0] == 0;\n}\n' % ', '.join('(fn_t)%s' % n for n in syms))
    exe = tmp_path / 'caller'
    libdir = os.path.dirname(uh_lib_path)
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'), str(src),
                        '-L', libdir, '-l:' + os.path.basename(uh_lib_path), '-Wl,-rpath,' + libdir,
                        '-Wl,--unresolved-symbols=ignore-in-shared-libs', '-o', str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # (not executed: loading libamdhip64 on a GPU-less host is the binding test's business)


def test_library_exports_every_declared_symbol(uh_lib_path):
    lib = C.CDLL(uh_lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), 'missing export: ' + name


def test_binding_covers_header(uh_lib_path):
    from unsuperviseddeephomographyral2018_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.uh_abi_version() == _lib.UH_ABI_VERSION


def test_argument_errors_without_gpu(uh_lib_path):
    """Argument validation happens before any HIP call, so it is testable on a GPU-less host."""
    from unsuperviseddeephomographyral2018_amd import _lib
    lib = _lib.load()
    one = C.c_void_p(16)     # never dereferenced: validation fails first
    assert lib.uh_warp_forward(None, one, one, None, 1, 8, 8, 3, 8, 8, None) == -1          # UH_E_NULL
    assert lib.uh_warp_forward(one, one, one, None, 0, 8, 8, 3, 8, 8, None) == -2           # UH_E_SHAPE
    assert lib.uh_warp_forward(one, one, one, None, 1, 8, 8, 5, 8, 8, None) == -3           # UH_E_CHANNELS
    assert lib.uh_warp_forward(one, one, one, None, 1, 1 << 14, 1 << 14, 3, 8, 8, None) == -5  # UH_E_TOO_LARGE
    assert lib.uh_warp_backward(one, one, one, one, None, None, 0, 1, 8, 8, 3, 8, 8, None) == -4  # UH_E_WORKSPACE
    assert lib.uh_dlt_forward(one, one, None, None, None, None, 4, 0, None) == -1
    assert lib.uh_dlt_forward(one, one, one, one, None, None, 4, 0, None) == -1             # theta needs M
    assert lib.uh_dlt_backward(one, one, one, None, None, None, None, one, 4, 0, None) == -1  # dH xor dtheta
    assert lib.uh_dlt_backward(one, one, one, one, one, None, None, one, 4, 0, None) == -1
    assert lib.uh_l1_loss_forward(one, one, one, None, 0, 16, None) == -4
    assert lib.uh_warp_patch_l1_fwdbwd(one, one, one, one, one, one, None, None, 0, 1, 8, 8, 3, 16, None) == -4
    assert lib.uh_warp_backward_workspace_bytes(64, 240, 320, 3, 240, 320) == 64 * 5 * 15 * 9 * 4
    assert lib.uh_warp_backward_workspace_bytes(0, 240, 320, 3, 240, 320) == 0
    assert b'UH_E_WORKSPACE' in lib.uh_error_string(-4)
    assert lib.uh_kernel_name(2) == b'warp_forward'
    assert lib.uh_kernel_name(16) == b'patch_loss_backward' and lib.uh_kernel_name(17) == b'?'
    # round 5: the prefetch family (uh_prefetch, uh_prefetch_async, uh_prefetch_join, uh_dlt_forward_prefetch) is gone
    for gone in ('uh_prefetch', 'uh_prefetch_async', 'uh_prefetch_join', 'uh_dlt_forward_prefetch'):
        assert not hasattr(lib, gone), gone
    assert lib.uh_dlt_zeroed_pairs(None, 0, None) == -1                                       # count is required
    assert b'UH_E_CAPTURING' in lib.uh_error_string(-6)
    # entry points added after the first slice
    assert lib.uh_warp_forward_literal(one, one, None, 1, 8, 8, 3, 8, 8, None) == -1
    assert lib.uh_patch_losses_forward(one, one, one, None, one, one, 1 << 20, 2, 16, None) == -1   # h4p xor gt
    assert lib.uh_patch_losses_forward(one, one, None, None, one, one, 1 << 20, 2, 2, None) == -2   # P < 3 (3x3 SSIM window)
    assert lib.uh_patch_losses_forward(one, one, None, None, one, None, 0, 2, 16, None) == -4
    assert lib.uh_patch_losses_workspace_bytes(64, 128) == 64 * 16 * 7 * 4
    assert lib.uh_patch_loss_backward(7, one, one, one, one, one, 2, 16, None) == -2             # unknown loss kind
    assert lib.uh_patch_loss_backward(1, one, one, None, one, one, 2, 16, None) == -1
    assert lib.uh_warp_patch_backward(one, one, one, one, one, None, 0, 2, 16, 16, 3, 64, None) == -4
    assert lib.uh_warp_patch_backward_workspace_bytes(64, 240, 320, 3) == 64 * 5 * 15 * (9 * 4 + 4)
    assert lib.uh_gather_patch_losses_forward(one, one, one, None, None, one, one, one, 1 << 20, 2, 16, 16, 3, 2, None) == -2   # P < 3
    assert lib.uh_gather_patch_losses_forward(one, one, one, one, None, one, one, one, 1 << 20, 2, 16, 16, 3, 8, None) == -1   # h4p xor gt
    assert lib.uh_gather_patch_losses_forward(one, one, one, None, None, one, one, one, 1 << 20, 2, 16, 16, 7, 8, None) == -3
    assert lib.uh_gather_patch_losses_forward(one, one, one, None, None, one, one, None, 0, 2, 16, 16, 3, 8, None) == -4
    assert lib.uh_warp_patch_loss_backward(1, one, one, one, one, one, None, one, one, one, 1 << 20, 2, 16, 16, 3, 64, None) == -2   # SSIM: stencil
    assert lib.uh_warp_patch_loss_backward(2, one, one, one, one, None, None, one, one, one, 1 << 20, 2, 16, 16, 3, 64, None) == -1
    assert lib.uh_warp_patch_loss_backward(2, one, one, one, one, one, None, one, one, None, 0, 2, 16, 16, 3, 64, None) == -4
    assert lib.uh_tail_create(C.byref(C.c_void_p()), 4, 60, 80, 3, 2, 0) == -2                       # un-fused tail: P >= 3
    args13 = [one] * 13
    assert lib.uh_prepare_inputs(*args13, 0, 8, 8, 4, None) == -2
    assert lib.uh_prepare_inputs(*args13, 2, 8, 8, 16, None) == -2                                  # patch larger than frame
    assert lib.uh_prepare_inputs(None, *args13[1:], 2, 8, 8, 4, None) == -1
    assert lib.uh_bias_relu_forward(one, one, one, 100, 6, None) == -3                                    # C % 4 != 0
    assert lib.uh_bias_relu_forward(one, one, one, 100, 96, None) == -3                                   # 1024 % C != 0
    assert lib.uh_bias_relu_forward(one, None, one, 100, 64, None) == -1
    assert lib.uh_bias_relu_backward(one, one, one, one, None, 0, 100, 64, None) == -4
    assert lib.uh_bias_relu_pool_forward(one, one, one, one, 2, 7, 8, 64, None) == -2                     # odd height
    assert lib.uh_bias_relu_pool_backward_workspace_bytes(2, 7, 8, 64) == 0
    plan = C.c_void_p()
    assert lib.uh_tail_create(C.byref(plan), 4, 60, 80, 3, 128, 0) == -2                             # patch larger than frame
    assert lib.uh_tail_create(C.byref(plan), 4, 60, 80, 5, 32, 0) == -3
    assert lib.uh_tail_create(C.byref(plan), 4, 60, 80, 3, 32, _lib.UH_TAIL_GRAPH) == 0 and plan.value
    nb_full = lib.uh_tail_workspace_bytes(plan)
    # one full frame (warped); the gradient frame is gone: the backward takes dPred directly (uh_warp_patch_backward)
    assert 4 * 60 * 80 * 3 * 4 <= nb_full < 2 * 4 * 60 * 80 * 3 * 4 and lib.uh_tail_warped_offset(plan) % 256 == 0
    assert lib.uh_tail_run(plan, *([one] * 11), one, nb_full - 1, None) == -4
    assert lib.uh_tail_run(plan, None, *([one] * 10), one, nb_full, None) == -1
    lib.uh_tail_destroy(plan)
    plan2 = C.c_void_p()
    assert lib.uh_tail_create(C.byref(plan2), 4, 60, 80, 3, 32, _lib.UH_TAIL_FUSED_PATCH) == 0
    assert lib.uh_tail_workspace_bytes(plan2) < 64 * 1024 and lib.uh_tail_warped_offset(plan2) == C.c_size_t(-1).value
    lib.uh_tail_destroy(plan2)


def test_product_path_refuses_cpu_tensors(uh_lib_path):
    torch = pytest.importorskip('torch')
    from unsuperviseddeephomographyral2018_amd import ops, _lib
    U = torch.zeros(1, 8, 8, 3)
    th = torch.eye(3).reshape(1, 9)
    with pytest.raises(_lib.UHError, match='no CPU fallback'):
        ops.transformer(U, th, (8, 8))
    with pytest.raises(_lib.UHError, match='no CPU fallback'):
        ops.solve_dlt(torch.zeros(1, 8), torch.zeros(1, 8))
    idx = torch.zeros(1, 16, dtype=torch.int32)
    with pytest.raises(_lib.UHError):
        ops.warp_gather_losses(U, th, idx, 4, torch.zeros(1, 4, 4, 1), train='l1_loss')
    with pytest.raises(_lib.UHError, match='no gradient kernel'):
        ops.warp_gather_losses(U, th, idx, 4, torch.zeros(1, 4, 4, 1), train='h_loss')


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, 'unsuperviseddeephomographyral2018_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: with the shared library absent, loading (and therefore every op) raises, it never computes."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from unsuperviseddeephomographyral2018_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.UHError as e:\n    print('RAISED', 'no CPU fallback' in str(e).replace('There is no', 'no'))\n"
            "else:\n    print('LOADED')\n") % ROOT
    env = dict(os.environ, UH_LIB_PATH=str(tmp_path / 'libuh_missing.so'))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert 'RAISED True' in out.stdout, out.stdout + out.stderr
