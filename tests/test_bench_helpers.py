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
         'warp_backward_kernel<3, false, true, false>@38400': {'hbm_bytes_per_launch': 747, 'launches_seen': 60},
         'prefetch_kernel@256': {'hbm_bytes_per_launch': 59, 'launches_seen': 8}}
    assert bench.traffic_lookup(m, 'warp_forward_kernel', 4800)['hbm_bytes_per_launch'] == 106
    assert bench.traffic_lookup(m, 'warp_forward_kernel', 38400)['hbm_bytes_per_launch'] == 751
    assert bench.traffic_lookup(m, 'warp_backward_kernel', 38400)['hbm_bytes_per_launch'] == 747
    assert bench.traffic_lookup(m, 'warp_backward_kernel', 1600) is None            # PATCH mode: not the dense kernel
    assert bench.traffic_lookup(m, 'prefetch_kernel', 256)['hbm_bytes_per_launch'] == 59
    assert bench.traffic_lookup(m, 'warp_forward_kernel', 123) is None
    assert bench.traffic_lookup({'error': 'rocprofv3 not found'}, 'warp_forward_kernel', 4800) is None
    assert bench.traffic_lookup(None, 'warp_forward_kernel', 4800) is None


def test_committed_traffic_is_flagged_stale_when_the_kernel_sources_moved(bench, monkeypatch):
    """profiles/traffic_rNN.json carries build._fingerprint() of the sources it was measured on; a different loaded library =>
    `stale`.  (Files from before round 4 carry no fingerprint and always count as stale.)"""
    f = os.path.join(ROOT, 'profiles', 'traffic_r04.json')
    tr = json.load(open(f))
    assert '_fingerprint' in tr and '_provenance' in tr
    key = 'warp_forward_B64_240x320'
    monkeypatch.setattr(bench, 'library_fingerprint', lambda: tr['_fingerprint'])
    got = bench.committed_traffic(key)
    assert got['file'] == 'profiles/traffic_r04.json' and got['stale'] is False
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


def test_timed_steps_stats_blocks(tmp_path):
    """tools/timed_steps_stats.py on a synthetic kernel trace: W warm-up + K timed steps, a stand-alone DLT solve (no backward:
    not a step), the 10-step round-1-law replay, then 33 steps with the frame prefetch and 23 without."""
    rows, t = [], [0]

    def k(name, dur):
        rows.append({'Kernel_Name': name, 'Start_Timestamp': t[0], 'End_Timestamp': t[0] + dur}); t[0] += dur + 100

    def step(fwd_ns, prefetch=False):
        k('void uh::dlt_forward_kernel<float>(...)', 5000)
        if prefetch:
            k('uh::prefetch_kernel(...)', 15000)
        k('void uh::warp_forward_kernel<3, false, true>(...)', fwd_ns)
        k('void uh::dlt_backward_kernel<float>(...)', 6000)
    W, K = 2, 5
    for _ in range(W):
        k('naive_conv_find_trial', 300000000); step(40000)
    for _ in range(K):
        step(27000)
    k('void uh::dlt_forward_kernel<float>(...)', 5000)                 # th_last: a solve without a backward
    for _ in range(10):
        step(23000)
    k('void uh::dlt_forward_kernel<float>(...)', 5000)
    for _ in range(33):
        step(18700, prefetch=True)
    for _ in range(23):
        step(27700)
    f = tmp_path / 'kernel_trace.csv'
    with open(f, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        w.writeheader(); w.writerows(rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'timed_steps_stats.py'), str(f), str(W), str(K)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    for row in csv.DictReader(io.StringIO(r.stdout)):
        if 'warp_forward' in row['Name']:
            got[row['Block'][:12]] = (int(row['Calls']), float(row['AverageNs']))
        assert 'naive_conv' not in row['Name']                         # find-mode trial kernels never enter
    assert got['timed steps '] == (K, 27000.0)
    assert got['replay under'] == (8, 23000.0)
    assert got['mid-training'][1] in (18700.0, 27700.0) and len(got) >= 3
    blocks = {row['Block'] for row in csv.DictReader(io.StringIO(r.stdout))}
    assert any('WITH the frame prefetch' in b for b in blocks) and any('prefetch off again' in b for b in blocks)


def test_regressor_flops_match_the_survey(bench):
    """2.52 GFLOP forward per pair at P = 128 (block1 1.246, block2 0.604, block3 0.453, block4 0.151, fc 0.067)."""
    assert abs(bench.regressor_flops(128) / 1e9 - 2.52) < 0.005
    assert bench.regressor_flops(64) < bench.regressor_flops(128) / 3.9


def test_committed_traffic_file_was_measured_on_the_current_kernel_sources():
    """The fallback for roofline.traffic must not go stale silently: profiles/traffic_r04.json stores build._fingerprint() of
    the sources it was measured on.  Editing csrc/ or include/ (even a comment: the header is part of the build stamp) makes
    this fail until `tools/gpu_session.sh rNN traffic` has been re-run and its file committed -- or the name of the newest
    file in bench.TRAFFIC_FILES bumped."""
    from unsuperviseddeephomographyral2018_amd import build
    tr = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_r04.json')))
    assert tr['_fingerprint'] == build._fingerprint()
