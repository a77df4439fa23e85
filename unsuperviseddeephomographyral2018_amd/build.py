"""Build the gfx950 C-ABI shared library IN-TREE (lib/libuh_hotpath.so).

hipcc cross-compiles without a GPU.  The built .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  `-ffp-contract=off` is part of the numerics contract: the forward
kernels must round once per written operation, like the un-fused TF-CPU graph.
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libuh_hotpath.so')
SOURCES = ['uh_dlt.hip', 'uh_warp.hip', 'uh_misc.hip', 'uh_patch.hip', 'uh_losses.hip', 'uh_inputs.hip', 'uh_tail.hip', 'uh_epilogue.hip']
HEADERS = ['uh_device.h', 'uh_host.h', os.path.join('..', '..', 'include', 'uh_hotpath.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-fPIC', '-shared',
         '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def _fingerprint():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def find_hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: cannot build libuh_hotpath.so')


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip -> lib/libuh_hotpath.so.  Returns the library path."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + '.sha256'
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return LIB
    cmd = [find_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    os.replace(LIB + '.tmp', LIB)
    with open(stamp, 'w') as fh:
        fh.write(fp)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
