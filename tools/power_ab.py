#!/usr/bin/env python
"""Time AND power of one warp kernel per library build (developer tool; VERDICT r3 item 3: the dense backward runs at the
board's power limit, so an experiment on it has to be read in joules per launch, not only in microseconds).

    python tools/power_ab.py --libs shipped,sg6,sg4 --kernel bwd --seconds 4 --reps 2

For every library (the shipped one, or lib/variants/libuh_<name>.so) and repetition a CHILD process loops the kernel at
BASELINE configs[3] (batch 128, 480x640, C=3, rho=64 law) for `--seconds`, first half with the in-library dispatch events on
(kernel duration), second half without (steady power); the parent samples `rocm-smi --showpower --showclocks --json`
meanwhile.  One JSON line per run: lib, kernel, avg_us, socket power and sclk (median of the samples taken while the child
looped), energy per launch = power x duration.
"""
import argparse
import json
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, 'unsuperviseddeephomographyral2018_amd', 'lib', 'variants')


def child(args):
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    from unsuperviseddeephomographyral2018_amd import _lib, ops, synthetic
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, H, W, P, rho = args.batch, args.h, args.w, 128, args.rho
    b = synthetic.make_batch(B, H, W, P, rho, seed=7, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    pred = b['gt'] + 2.0 * torch.randn(B, 8, generator=g, device=dev)
    _, theta = ops.solve_dlt(b['pts1'], pred, img_w=W, img_h=H)
    theta = theta.detach().contiguous()
    U = b['I_aug']
    dOut = torch.randn(B, H, W, 3, generator=g, device=dev)
    out = torch.empty_like(U)
    dT = torch.empty(B, 9, device=dev)
    nb = lib.uh_warp_backward_workspace_bytes(B, H, W, 3, H, W)
    ws = torch.empty(nb // 4, device=dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    if args.kernel == 'fwd':
        fn = lambda: lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, st())
        key = 'warp_forward'
    elif args.kernel == 'bwd':
        fn = lambda: lib.uh_warp_backward(p(U), p(theta), p(dOut), p(dT), None, p(ws), nb, B, H, W, 3, H, W, st())
        key = 'warp_backward'
    else:
        fn = lambda: out.copy_(U)
        key = None
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    print('LOOP_START', flush=True)
    t_end = time.time() + args.seconds / 2
    if key:
        _lib.profile_enable(True, only=(key,))
    n = 0
    while time.time() < t_end:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    us = None
    if key:
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        us = prof[key][0] / max(prof[key][1], 1) * 1e3
    t0 = time.time()
    t_end = t0 + args.seconds / 2
    m = 0
    while time.time() < t_end:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        m += 50
    wall_us = (time.time() - t0) / max(m, 1) * 1e6
    print('RESULT ' + json.dumps({'avg_us': round(us, 2) if us else None, 'loop_us_per_launch': round(wall_us, 2),
                                  'launches': n + m, 'dtheta_checksum': float(dT.double().abs().sum()) if key == 'warp_backward' else None}),
          flush=True)


def smi():
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
    except Exception:
        return None, None
    card = d.get('card0') or next(iter(d.values()), {})
    power = sclk = None
    for k, v in card.items():
        if 'ower' in k and power is None:
            m = re.search(r'[-+]?\d+(\.\d+)?', str(v))
            power = float(m.group(0)) if m else None
        if 'sclk' in k and sclk is None:
            m = re.search(r'(\d+)\s*[Mm][Hh]z', str(v))
            sclk = float(m.group(1)) if m else None
    return power, sclk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--libs', default='shipped')
    ap.add_argument('--kernel', default='bwd', choices=['fwd', 'bwd', 'copy'])
    ap.add_argument('--seconds', type=float, default=4.0)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--h', type=int, default=480)
    ap.add_argument('--w', type=int, default=640)
    ap.add_argument('--rho', type=int, default=64)
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child(args)
    for rep in range(args.reps):
        for name in args.libs.split(','):
            env = dict(os.environ)
            if name != 'shipped':
                env['UH_LIB_PATH'] = os.path.join(VDIR, 'libuh_%s.so' % name)
            cmd = [sys.executable, os.path.abspath(__file__), '--child', '--kernel', args.kernel, '--seconds', str(args.seconds),
                   '--batch', str(args.batch), '--h', str(args.h), '--w', str(args.w), '--rho', str(args.rho)]
            pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
            samples, res, started = [], None, False
            import threading
            lines = []

            def rd():
                for ln in pr.stdout:
                    lines.append(ln)
            th = threading.Thread(target=rd, daemon=True); th.start()
            while pr.poll() is None:
                if any(l.startswith('LOOP_START') for l in lines):
                    started = True
                if started:
                    pw, ck = smi()
                    if pw is not None:
                        samples.append((pw, ck))
                else:
                    time.sleep(0.2)
            th.join(timeout=5)
            for ln in lines:
                if ln.startswith('RESULT '):
                    res = json.loads(ln[7:])
            samples = samples[1:-1] if len(samples) > 4 else samples      # drop the ramp at either end
            pw = statistics.median([s[0] for s in samples]) if samples else None
            ck = statistics.median([s[1] for s in samples if s[1]]) if any(s[1] for s in samples) else None
            out = {'lib': name, 'kernel': args.kernel, 'rep': rep, 'power_W': pw, 'sclk_MHz': ck, 'smi_samples': len(samples)}
            out.update(res or {'error': 'child gave no result', 'rc': pr.returncode})
            if pw and res and res.get('avg_us'):
                out['mJ_per_launch'] = round(pw * res['avg_us'] * 1e-3, 2)
            print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
