"""Shared by tools/timed_steps_stats.py and tools/step_breakdown.py: cut a rocprofv3 --kernel-trace CSV of bench.py into
training steps, and stamp the outputs with the source fingerprint of the library the trace was taken on."""
import csv
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fingerprint():
    """build._fingerprint() of the kernel sources: the stamp hipcc's output was saved with (lib/libuh_hotpath.so.sha256 -- on
    the GPU box that is the library the traced process loaded), else computed from the sources."""
    stamp = os.path.join(ROOT, 'unsuperviseddeephomographyral2018_amd', 'lib', 'libuh_hotpath.so.sha256')
    try:
        return open(stamp).read().strip()
    except OSError:
        import sys
        sys.path.insert(0, ROOT)
        from unsuperviseddeephomographyral2018_amd import build
        return build._fingerprint()


def read_fingerprint(path):
    """The `# _fingerprint: <sha256>` line an evidence file under profiles/ starts with; None when it carries none."""
    with open(path) as fh:
        for line in fh:
            if line.startswith('# _fingerprint:'):
                return line.split(':', 1)[1].split()[0]
            if not line.startswith('#'):
                break
    return None


def load_steps(trace_csv):
    """-> (rows sorted by start time, [(first_row, end_row) per training step]).  A step is the span from one
    uh::dlt_forward_kernel<float> launch to the next that also holds a uh::dlt_backward launch (the stand-alone DLT solves
    bench.py makes for its statistics hold none)."""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'dlt_forward_kernel<float>' in r['Kernel_Name']]
    spans = []
    for a, b in zip(marks, marks[1:] + [len(rows)]):
        if any('dlt_backward_kernel' in rows[i]['Kernel_Name'] for i in range(a, b)):
            spans.append((a, b))
    return rows, spans
