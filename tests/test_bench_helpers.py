"""CPU: the measurement plumbing of bench.py and tools/ that the GPU runs rely on -- traffic bookkeeping (this run's PMC passes vs
the committed fallback and its staleness flag), the per-block kernel statistics of the timed steps, the committed
reference-schedule result.  No GPU, no rocprofv3."""
import csv
import importlib.util
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('uh_bench_helpers', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_lookup_tells_shapes_and_instantiations_apart(bench):
    """The in-step forward (4 800 blocks) and the config-4 forward (38 400) share a kernel name; the dense backward must not be
    confused with the PATCH-mode instantiation the train step launches."""
    m = {'_how': 'x', '_seconds': 1.0,
         'warp_forward_kernel<3, false, true>@4800': {'hbm_bytes_per_launch': 106, 'launches_seen': 9},
         'warp_forward_kernel<3, false, true>@38400': {'hbm_bytes_per_launch': 751, 'launches_seen': 61},
         'warp_backward_kernel<3, false, true, true>@1600': {'hbm_bytes_per_launch': 30, 'launches_seen': 8},
         'warp_backward_kernel<3, false, true, false>@38400': {'hbm_bytes_per_launch': 747, 'launches_seen': 60}}
    assert bench.traffic_lookup(m, 'warp_forward_kernel', 4800)['hbm_bytes_per_launch'] == 106
    assert bench.traffic_lookup(m, 'warp_forward_kernel', 38400)['hbm_bytes_per_launch'] == 751
    assert bench.traffic_lookup(m, 'warp_backward_kernel', 38400)['hbm_bytes_per_launch'] == 747
    assert bench.traffic_lookup(m, 'warp_backward_kernel', 1600) is None            # PATCH mode: not the dense kernel
    assert bench.traffic_lookup(m, 'warp_forward_kernel', 123) is None
    assert bench.traffic_lookup({'error': 'rocprofv3 not found'}, 'warp_forward_kernel', 4800) is None
    assert bench.traffic_lookup(None, 'warp_forward_kernel', 4800) is None


def test_committed_traffic_is_flagged_stale_when_the_kernel_sources_moved(bench, monkeypatch):
    """profiles/traffic_rNN.json carries build._fingerprint() of the sources it was measured on; a different loaded library =>
    `stale`.  (Files from before round 4 carry no fingerprint and always count as stale.)"""
    newest = next(n for n in bench.TRAFFIC_FILES if os.path.exists(os.path.join(ROOT, 'profiles', n)))
    f = os.path.join(ROOT, 'profiles', newest)
    tr = json.load(open(f))
    assert '_fingerprint' in tr and '_provenance' in tr
    key = 'warp_forward_B64_240x320'
    monkeypatch.setattr(bench, 'library_fingerprint', lambda: tr['_fingerprint'])
    got = bench.committed_traffic(key)
    assert got['file'] == 'profiles/' + newest and got['stale'] is False
    assert got['hbm_bytes_per_launch'] == tr[key]['hbm_bytes_per_launch']
    # under the mid-training law a fifth of U is never sampled: the file must hold the TIMED law's bytes (0.90 x algorithmic),
    # not the near-identity replay's 0.997 x that rounds 1-3 quoted
    alg = 2 * 64 * 240 * 320 * 3 * 4
    assert 0.85 < got['hbm_bytes_per_launch'] / alg < 0.95
    monkeypatch.setattr(bench, 'library_fingerprint', lambda: 'somethingelse')
    assert bench.committed_traffic(key)['stale'] is True
    monkeypatch.setattr(bench, 'TRAFFIC_FILES', ('traffic_r03.json',))
    assert bench.committed_traffic(key)['stale'] is True                             # no fingerprint recorded
    assert bench.committed_traffic('no_such_kernel') is None


def test_measure_traffic_reports_a_missing_profiler_instead_of_raising(bench, monkeypatch):
    import shutil
    monkeypatch.setattr(shutil, 'which', lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, 'exists', lambda p: False if 'rocprofv3' in str(p) else real_exists(p))
    out = bench.measure_traffic(type('A', (), {})())
    assert out == {'error': 'rocprofv3 not found'}
    # ... and never starts a profiler inside a profiler
    monkeypatch.setattr(os.path, 'exists', real_exists)
    monkeypatch.setattr(shutil, 'which', lambda name: sys.executable)
    monkeypatch.setenv('ROCPROFILER_SOMETHING', '1')
    assert 'running under a profiler' in bench.measure_traffic(type('A', (), {})())['error']


def test_reference_schedule_result_is_read_from_the_committed_log(bench):
    r = bench.reference_schedule_result()
    assert r['train_steps'] == 150000 and r['file'].startswith('profiles/') and 'not measured in this run' in r['note']
    assert 3.0 < r['mean_corner_error_px'] < 8.0 and 0.0 <= r['fail_percent'] < 1.0
    # round 5: the same schedule FROM JPEG FILES through Dataloader + uh_prepare_inputs, tested with and without the reference's
    # disjoint test augmentation -- both rows are carried, neither is measured in the bench run
    d = r['from_jpeg_files_with_augmentation']
    assert d['file'] == 'profiles/r06_train_from_disk_reference_schedule.txt' and 'not measured in this run' in d['note']
    rows = {row['test_do_augment']: row for row in d['results']}
    assert set(rows) == {0.0, 0.5} and all(row['steps'] == 150000 and row['files'] == 'jpg' for row in rows.values())
    assert rows[0.0]['mean_corner_error_px'] < 6.0 and rows[0.0]['fail_percent'] < 0.5           # the producers are sound ...
    assert rows[0.5]['mean_corner_error_px'] > 2 * rows[0.0]['mean_corner_error_px']           # ... the disjoint test noise is what hurts


def _synthetic_trace(tmp_path, W, K):
    rows, t = [], [0]

    def k(name, dur):
        rows.append({'Kernel_Name': name, 'Start_Timestamp': t[0], 'End_Timestamp': t[0] + dur}); t[0] += dur + 100

    def step(fwd_ns):
        k('void uh::dlt_forward_kernel<float>(...)', 5000)
        k('void uh::warp_forward_kernel<3, false, true>(...)', fwd_ns)
        k('void uh::dlt_backward_kernel<float>(...)', 6000)
    for _ in range(W):
        k('naive_conv_find_trial', 300000000); step(40000)
    for _ in range(K):
        step(27000)
    k('void uh::dlt_forward_kernel<float>(...)', 5000)                 # th_last: a solve without a backward
    for _ in range(10):
        step(23000)
    k('void uh::dlt_forward_kernel<float>(...)', 5000)                 # the warp-only points: one more stand-alone solve
    f = tmp_path / 'kernel_trace.csv'
    with open(f, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        w.writeheader(); w.writerows(rows)
    return f


def test_timed_steps_stats_blocks(tmp_path):
    """tools/timed_steps_stats.py on a synthetic kernel trace: W warm-up + K timed steps, a stand-alone DLT solve (no backward:
    not a step), then the 10-step round-1-law replay.  The output starts with the source fingerprint of the library."""
    from unsuperviseddeephomographyral2018_amd import build
    W, K = 2, 5
    f = _synthetic_trace(tmp_path, W, K)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'timed_steps_stats.py'), str(f), str(W), str(K)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    head, body = r.stdout.split('\n', 1)
    assert head.startswith('# _fingerprint: ' + build._fingerprint())
    got = {}
    for row in csv.DictReader(io.StringIO(body)):
        if 'warp_forward' in row['Name']:
            got[row['Block'][:12]] = (int(row['Calls']), float(row['AverageNs']))
        assert 'naive_conv' not in row['Name']                         # find-mode trial kernels never enter
    assert got == {'timed steps ': (K, 27000.0), 'replay under': (8, 23000.0)}
    # ... and the per-step breakdown of the same K timed steps (not the replay, not the warm-up)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'step_breakdown.py'), str(f), str(W), str(K)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith('# _fingerprint: ' + build._fingerprint())
    assert 'steps averaged: %d' % K in lines[1] and 'sum of kernel durations per step 38.0 us' in lines[1]
    assert 'wall per step 38.3 us' in lines[1]                         # 3 kernels + 3 x 100 ns gaps: not the read-back pause after the timed region
    assert any('27.0 us/step' in l and 'warp_forward' in l for l in lines) and not any('naive_conv' in l for l in lines)


def test_regressor_flops_match_the_survey(bench):
    """2.52 GFLOP forward per pair at P = 128 (block1 1.246, block2 0.604, block3 0.453, block4 0.151, fc 0.067)."""
    assert abs(bench.regressor_flops(128) / 1e9 - 2.52) < 0.005
    assert bench.regressor_flops(64) < bench.regressor_flops(128) / 3.9


EVIDENCE_ROUND = 'r06'        # the round whose profiles/ evidence must have been taken on the kernel sources in the tree


def test_committed_evidence_was_measured_on_the_current_kernel_sources():
    """The evidence the documents quote must not go stale silently: profiles/traffic_rNN.json (the fallback for
    roofline.traffic), rNN_bench_kernel_stats_timed_steps.csv and rNN_step_breakdown.txt each store build._fingerprint() of the
    sources they were measured on.  Editing csrc/ or include/ (even a comment: the header is part of the build stamp) makes
    this fail until `tools/gpu_session.sh rNN rocprof traffic` has been re-run and its files committed."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from _trace_steps import read_fingerprint
    from unsuperviseddeephomographyral2018_amd import build
    fp = build._fingerprint()
    tr = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_%s.json' % EVIDENCE_ROUND)))
    assert tr['_fingerprint'] == fp
    for name in ('%s_bench_kernel_stats_timed_steps.csv' % EVIDENCE_ROUND, '%s_step_breakdown.txt' % EVIDENCE_ROUND):
        assert read_fingerprint(os.path.join(ROOT, 'profiles', name)) == fp, name


def test_bench_source_holds_no_frozen_measurements():
    """the bench line carries what the run measured and file pointers, not prose with numbers somebody measured
    once ("fc1 forward 142 -> 55 us", "the hot path is 1.5 % of the step").  Checked on the SOURCE: no string literal of bench.py
    outside docstrings holds a literal number followed by a measurement unit -- numbers reach the line through format specifiers
    fed by variables.  Allowed: the constants of a definition (the theta law N(0, 2 px))."""
    import ast
    import re
    src = open(os.path.join(ROOT, 'bench.py')).read()
    tree = ast.parse(src)
    docstrings = set()
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.Module, ast.ClassDef)) and n.body and isinstance(n.body[0], ast.Expr) \
                and isinstance(n.body[0].value, ast.Constant) and isinstance(n.body[0].value.value, str):
            docstrings.add(id(n.body[0].value))
    unit = re.compile(r'(?<![%\w.])(\d+(?:[ .]\d+)*)\s*(us|\u00b5s|ms|%|px|pairs/s|GB/s|TB/s|TFLOP/s)(?![\w/])')
    allowed = {'2 px', '2px'}
    found = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Constant) and isinstance(n.value, str) and id(n) not in docstrings:
            found += [(n.lineno, m.group(0)) for m in unit.finditer(n.value) if m.group(0) not in allowed]
    assert not found, 'literal measurements in bench.py strings (use a variable or a profiles/ pointer): %r' % found
    # every profiles/ file the line points at exists
    for f in set(re.findall(r'profiles/[\w./-]+\.(?:jsonl|json|txt|csv)', src)):
        if '%' in f or 'rNN' in f or f.startswith('profiles/r06_'):             # round-6 evidence is written by the GPU session of this round
            continue
        assert os.path.exists(os.path.join(ROOT, f)), f
