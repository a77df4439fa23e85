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


def kernel_resources(force=False):
    """Register / LDS / scratch budget of EVERY kernel of the library, from the code-object metadata hipcc emits
    (`-S --cuda-device-only`: device assembly only, ~10 s for uh_warp.hip, the files in parallel), cached beside the library
    under the same source fingerprint.  -> {demangled kernel name: {'file', 'vgpr', 'agpr', 'sgpr', 'vgpr_spill', 'sgpr_spill',
    'scratch_bytes', 'lds_bytes', 'dynamic_stack', 'max_threads'}}.  tests/test_kernel_resources.py holds the budgets."""
    import json
    import re
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    import yaml
    os.makedirs(LIBDIR, exist_ok=True)
    cache = os.path.join(LIBDIR, 'kernel_resources.json')
    fp = _fingerprint()
    if not force and os.path.exists(cache):
        try:
            with open(cache) as fh:
                c = json.load(fh)
            if c.get('fingerprint') == fp:
                return c['kernels']
        except Exception:
            pass
    dev_flags = [f for f in FLAGS if f not in ('-shared', '-fPIC')] + ['-S', '--cuda-device-only', '-Wno-unused-command-line-argument']
    hipcc = find_hipcc()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        def one(src):
            asm = os.path.join(td, src + '.s')
            res = subprocess.run([hipcc] + dev_flags + [os.path.join(CSRC, src), '-o', asm], capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError('hipcc -S failed for %s:\n%s%s' % (src, res.stdout, res.stderr))
            with open(asm) as fh:
                text = fh.read()
            m = re.search(r'\.amdgpu_metadata\n---\n(.*?)\n\.\.\.\n', text, re.S)
            return src, (yaml.safe_load(m.group(1)).get('amdhsa.kernels', []) if m else [])
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            per_file = list(ex.map(one, SOURCES))
    names = [k['.name'] for _, ks in per_file for k in ks]
    filt = shutil.which('c++filt')
    dem = subprocess.run([filt] + names, capture_output=True, text=True).stdout.split('\n') if (filt and names) else names
    i = 0
    for src, ks in per_file:
        for k in ks:
            name = re.sub(r'\(.*$', '', dem[i]).replace('void ', '') or k['.name']
            i += 1
            # a key the toolchain did not emit is None (NOT 0): tests/test_kernel_resources.py must fail on it instead of reading
            # "no spills" out of a renamed metadata field
            out[name] = {'file': src, 'vgpr': k.get('.vgpr_count'), 'agpr': k.get('.agpr_count', 0), 'sgpr': k.get('.sgpr_count'),
                         'vgpr_spill': k.get('.vgpr_spill_count'), 'sgpr_spill': k.get('.sgpr_spill_count'),
                         'scratch_bytes': k.get('.private_segment_fixed_size'), 'lds_bytes': k.get('.group_segment_fixed_size'),
                         'dynamic_stack': bool(k.get('.uses_dynamic_stack', False)),
                         'max_threads': k.get('.max_flat_workgroup_size')}
    with open(cache, 'w') as fh:
        json.dump({'fingerprint': fp, 'kernels': out}, fh, indent=0, sort_keys=True)
    return out


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
