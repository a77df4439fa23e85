#!/usr/bin/env python
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing in tests/, smoke()
or bench.py reads /root/reference at run time -- they read the .npz files this script wrote.

Two kinds of vectors:

(1) REFERENCE-DERIVED (pin the oracle against the reference's own code)
  * ref_numpy_transformer.npz -- outputs of the reference's utils/numpy_spatial_transformer.py
    `_meshgrid` (:12-24) and `_interpolate` (:27-94), imported as-is with cv2/skimage/matplotlib
    stubbed in sys.modules (they are only used by its __main__ self-test), on gray images (its
    3-channel branch is broken under numpy>=2).  Includes the reference's own self-test homography
    H = [[2,.3,5],[.3,2,10],[1e-4,2e-4,1]] (:157).
  * ref_dlt_system.npz -- A and b of the Tensor-DLT built by literally evaluating the reference's
    formula (homography_model.py:223-238) with the reference's own Aux_M* constants, which are
    extracted from utils/utils.py:11-122 with `ast` (the module itself cannot be imported: it needs
    cv2, tensorflow and a TTY).

  * ref_numpy_transformer_rgb.npz (round 5) -- the same reference functions applied CHANNEL BY CHANNEL to a 3-channel image (the
    reference's own 3-channel branch, `np.expand_dims(wa, 2)` :88-91, raises under numpy >= 2; bilinear sampling acts on each
    channel independently, so three gray calls ARE its 3-channel semantics).  Pins the oracle's C = 3 path -- the shape the
    product runs -- to reference code instead of to the restatement alone.  `python make_golden.py rgb` writes only this file.

  * ref_find_percentile.npz (round 6) -- outputs of the reference's utils.find_percentile (utils/utils.py:655-672), extracted with
    `ast` like Aux_M* and executed as-is, on the reference's own test vector (test_find_percentile, :675) and on seeded vectors
    of several lengths.  `python make_golden.py percentile` writes only this file.
  * ref_cli_flags.json (round 6) -- every parser.add_argument of homography_CNN_synthetic.py:49-85 (flag, type name, nargs,
    choices, default where it is a literal) read with `ast`, plus the README's homography_CNN_synthetic.py command lines with
    their line numbers.  Data for tests/test_cli.py.  `python make_golden.py cli` writes only this file.

(2) ORACLE-DERIVED (freeze the f64/f32 oracle so later edits cannot drift silently, and give the
    GPU tests fixed inputs+outputs that travel to the GPU box)
  * chain_small.npz -- full photometric chain fwd+bwd on a small seeded batch.
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/code'


def import_reference_numpy_transformer():
    for name in ('cv2', 'skimage', 'skimage.io', 'matplotlib', 'matplotlib.pyplot', 'pdb'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['skimage'].io = sys.modules['skimage.io']
    sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'ref_numpy_spatial_transformer', os.path.join(REF, 'utils', 'numpy_spatial_transformer.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def extract_aux_matrices():
    src = open(os.path.join(REF, 'utils', 'utils.py')).read()
    tree = ast.parse(src)
    ns = {'np': np}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 \
                and isinstance(node.targets[0], ast.Name) and node.targets[0].id.startswith('Aux_M'):
            exec(compile(ast.Module([node], []), 'utils.py', 'exec'), ns)
    return {k: v for k, v in ns.items() if k.startswith('Aux_M')}


def reference_dlt_system(aux, pts1, h4p):
    """homography_model.py:171-238 evaluated with numpy in f32 (batched matmul, same formula)."""
    f = np.float32
    pts1_t = pts1.astype(f)[:, :, None]
    p2_t = (h4p.astype(f)[:, :, None] + pts1_t).astype(f)
    M = {k: v.astype(f)[None] for k, v in aux.items()}
    B = pts1.shape[0]
    A1 = M['Aux_M1'] @ pts1_t
    A2 = M['Aux_M2'] @ pts1_t
    A3 = np.tile(M['Aux_M3'], (B, 1, 1))
    A4 = M['Aux_M4'] @ pts1_t
    A5 = M['Aux_M5'] @ pts1_t
    A6 = np.tile(M['Aux_M6'], (B, 1, 1))
    A7 = (M['Aux_M71'] @ p2_t) * (M['Aux_M72'] @ pts1_t)
    A8 = (M['Aux_M71'] @ p2_t) * (M['Aux_M8'] @ pts1_t)
    cols = [a.reshape(-1, 8) for a in (A1, A2, A3, A4, A5, A6, A7, A8)]
    A = np.transpose(np.stack(cols, axis=1), (0, 2, 1))
    b = (M['Aux_Mb'] @ p2_t).reshape(-1, 8)
    return A.astype(f), b.astype(f)


def extract_find_percentile():
    """utils.find_percentile (utils/utils.py:655-672) as a callable: the FunctionDef node is compiled as-is (the module cannot
    be imported: cv2, tensorflow, a TTY); its print() calls are silenced."""
    src = open(os.path.join(REF, 'utils', 'utils.py')).read()
    ns = {'np': np, 'print': lambda *a, **k: None}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == 'find_percentile':
            exec(compile(ast.Module([node], []), 'utils.py', 'exec'), ns)
    return ns['find_percentile']


def main_percentile():
    fp = extract_find_percentile()
    rs = np.random.RandomState(655)
    out = {}
    x_ref = np.array([10, 1.10, 2, 3, 4, 5, 6, 6, 7, 8, 9], np.float64)          # test_find_percentile (:675)
    out['x_ref'] = x_ref; out['y_ref'] = np.asarray(fp(list(x_ref)), np.float64)
    for n in (3, 4, 10, 24, 100, 1001):                                          # 24 = 3 epochs x 8 test batches
        x = np.abs(rs.randn(n) * 6.0 + 5.0)
        out['x_%d' % n] = x; out['y_%d' % n] = np.asarray(fp(x), np.float64)
    x32 = (rs.rand(37) * 40).astype(np.float32)                                  # the trainer hands it float32 per-pair arrays too
    out['x_f32'] = x32; out['y_f32'] = np.asarray(fp(x32), np.float64)
    np.savez_compressed(os.path.join(HERE, 'ref_find_percentile.npz'), **out)
    print('ref_find_percentile.npz written to', HERE)


def main_cli():
    import json
    src = open(os.path.join(REF, 'homography_CNN_synthetic.py')).read()
    flags = []
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'add_argument' \
                and isinstance(node.func.value, ast.Name) and node.func.value.id == 'parser':
            kw = {k.arg: k.value for k in node.keywords}
            ent = {'flag': node.args[0].value, 'line': node.lineno,
                   'type': kw['type'].id if 'type' in kw else None,
                   'nargs': kw['nargs'].value if 'nargs' in kw else None,
                   'choices': ast.literal_eval(kw['choices']) if 'choices' in kw else None}
            try:
                ent['default'] = ast.literal_eval(kw['default']); ent['default_is_literal'] = True
            except (ValueError, KeyError):
                ent['default'] = kw['default'].id if isinstance(kw.get('default'), ast.Name) else None
                ent['default_is_literal'] = False
            flags.append(ent)
    flags.sort(key=lambda e: e['line'])
    cmds = []
    for i, line in enumerate(open('/root/reference/README.md').read().splitlines(), 1):
        if line.strip().startswith('python homography_CNN_synthetic.py'):
            cmds.append({'line': i, 'argv': line.strip().split()[2:]})
    with open(os.path.join(HERE, 'ref_cli_flags.json'), 'w') as f:
        json.dump({'source': 'code/homography_CNN_synthetic.py:49-85 + README.md', 'flags': flags, 'readme_commands': cmds}, f, indent=1)
    print('ref_cli_flags.json: %d flags, %d README commands' % (len(flags), len(cmds)))


def main_rgb():
    """ref_numpy_transformer_rgb.npz: the reference's _meshgrid / _interpolate, one call per channel of an RGB image, under
    the reference's self-test homography (:157), a mild perspective and a strong one whose samples leave the frame."""
    ref = import_reference_numpy_transformer()
    rs = np.random.RandomState(4321)
    H_, W_ = 40, 56
    img = rs.uniform(0, 255, size=(H_, W_, 3))                    # RGB, f64
    M = np.array([[W_ / 2.0, 0, W_ / 2.0], [0, H_ / 2.0, H_ / 2.0], [0, 0, 1.]]).astype(np.float32)
    Hs = [np.array([[2., 0.3, 5], [0.3, 2., 10.], [0.0001, 0.0002, 1.]], np.float32),   # :157
          np.array([[0.95, 0.08, -2.], [-0.06, 1.05, 3.], [3e-4, -2e-4, 1.]], np.float32),
          np.array([[1.4, -0.3, -25.], [0.25, 0.7, 12.], [3e-3, 1e-3, 1.]], np.float32)]
    thetas, outs = [], []
    for Hm in Hs:
        theta = np.dot(np.dot(np.linalg.inv(M), np.linalg.inv(Hm)), M)      # numpy_transformer() :136-141
        T = np.dot(theta, ref._meshgrid(H_, W_))
        xs, ys = T[0] / T[2], T[1] / T[2]
        chans = [ref._interpolate(np.ascontiguousarray(img[:, :, c]), xs, ys, [H_, W_]).reshape(H_, W_) for c in range(3)]
        thetas.append(theta); outs.append(np.stack(chans, 2))
    np.savez_compressed(os.path.join(HERE, 'ref_numpy_transformer_rgb.npz'), img=img, thetas=np.stack(thetas), outs=np.stack(outs))
    print('ref_numpy_transformer_rgb.npz written to', HERE)


def main():
    from oracle import hotpath_numpy as O
    rs = np.random.RandomState(1234)

    # ---- (1a) reference numpy transformer -------------------------------------------------
    ref = import_reference_numpy_transformer()
    out = {}
    H_, W_ = 48, 64
    img = rs.uniform(0, 255, size=(H_, W_))                       # gray, f64
    M = np.array([[W_ / 2.0, 0, W_ / 2.0], [0, H_ / 2.0, H_ / 2.0], [0, 0, 1.]]).astype(np.float32)
    Hs = [np.array([[2., 0.3, 5], [0.3, 2., 10.], [0.0001, 0.0002, 1.]], np.float32),   # :157
          np.eye(3, dtype=np.float32),
          np.array([[0.9, -0.1, 3.], [0.05, 1.1, -2.], [-4e-4, 3e-4, 1.]], np.float32),
          np.array([[1.3, 0.2, -20.], [-0.2, 0.8, 15.], [2e-3, -1e-3, 1.]], np.float32)]
    thetas, grids, outs = [], [], []
    for Hm in Hs:
        # same composition as numpy_transformer() :136-141 (theta = M^-1 H^-1 M)
        theta = np.dot(np.dot(np.linalg.inv(M), np.linalg.inv(Hm)), M)
        grid = ref._meshgrid(H_, W_)
        T = np.dot(theta, grid)
        xs, ys, ts = T[0], T[1], T[2]
        o = ref._interpolate(img, xs / ts, ys / ts, [H_, W_])
        thetas.append(theta); outs.append(o.reshape(H_, W_))
    out.update(img=img, thetas=np.stack(thetas), outs=np.stack(outs), grid=ref._meshgrid(H_, W_))
    np.savez_compressed(os.path.join(HERE, 'ref_numpy_transformer.npz'), **out)

    # ---- (1b) reference DLT system --------------------------------------------------------
    aux = extract_aux_matrices()
    assert len(aux) == 10, sorted(aux)
    B = 32
    x0 = rs.randint(45, 148, size=B); y0 = rs.randint(45, 68, size=B)
    pts1 = np.stack([x0, y0, x0 + 128, y0, x0 + 128, y0 + 128, x0, y0 + 128], 1).astype(np.float32)
    h4p = (rs.randint(-45, 46, size=(B, 8)) + rs.randn(B, 8)).astype(np.float32)
    A, b = reference_dlt_system(aux, pts1, h4p)
    np.savez_compressed(os.path.join(HERE, 'ref_dlt_system.npz'), pts1=pts1, h4p=h4p, A=A, b=b)

    # ---- (2) oracle-derived chain vectors -------------------------------------------------
    d = O.synthetic_batch(7, 6, H=60, W=80, P=32, rho=10)
    # include one strong homography whose far field leaves int32 range and one near-singular t
    d['pred_h4p'][0] = d['gt'][0]
    f32 = O.photometric_chain(d['I'], d['I2'], d['pts1'], d['pred_h4p'], d['patch_indices'], 32,
                              np.float32)
    bw = O.photometric_chain_backward(d['I'], d['I2'], d['pts1'], d['pred_h4p'],
                                      d['patch_indices'], 32)
    dOut = rs.randn(*d['I'].shape).astype(np.float32)
    dtheta_full = O.transformer_backward(d['I'], f32['theta'], dOut, (60, 80), np.float64)
    np.savez_compressed(
        os.path.join(HERE, 'chain_small.npz'),
        I=d['I'], I2=d['I2'], pts1=d['pts1'], gt=d['gt'], pred_h4p=d['pred_h4p'],
        patch_indices=d['patch_indices'],
        H32=f32['H'], theta32=f32['theta'], warped32=f32['warped'], pred32=f32['pred_I2'],
        loss32=np.float32(f32['l1_loss']),
        H64=bw['H'], theta64=bw['theta'], warped64=bw['warped'], loss64=np.float64(bw['l1_loss']),
        dtheta64=bw['dtheta'], dH64=bw['dH'], dh4p64=bw['dh4p'],
        dOut=dOut, dtheta_full64=dtheta_full)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    if sys.argv[1:] == ['rgb']:
        main_rgb()
    elif sys.argv[1:] == ['percentile']:
        main_percentile()
    elif sys.argv[1:] == ['cli']:
        main_cli()
    else:
        main()
        main_rgb()
        main_percentile()
        main_cli()
