#!/usr/bin/env python
"""What is the in-step forward's fixed cost made of?  (needs a library built with -DUH_WARP_TRACE:
`tools/variants.sh trace "-DUH_WARP_TRACE"` and UH_LIB_PATH=.../variants/libuh_trace.so.)

t(B) of uh_warp_forward at 240x320 on a cold frame is 9.5 us + 0.295 us * B (profiles/r04_cold_forward.jsonl): a third of
the batch-64 launch does not scale with the work.  This tool takes ONE traced launch at the in-step shape (batch 64, 240x320,
mid-training theta law, input evicted by a 1 GiB copy first) and reads every wave's entry / end on the chip-wide 100 MHz
counter (s_memrealtime, comparable across XCDs) plus its phase stamps on the shader clock (s_memtime), and the launch's
duration by dispatch events (uh_profile_*) -- the figure bench.py's roofline object reports.  From those:

  dispatch_overhead_us   event duration - (last wave end - first wave start): command processor launch + end-of-kernel release
  ramp_us                first wave start -> the chip holds 90 % of its plateau of resident waves
  first_store_us         first wave start -> the first wave anywhere has its data (path A: DMA landed; gather: loads back) --
                         before that no HBM store can be in flight
  tail_us                resident waves fall below 90 % of the plateau for good -> last wave end
  per XCD                first start, last end, waves, so that an XCD that starts late or finishes late is visible

Repeated --reps times (fresh eviction each time); prints one JSON line per repetition and a summary line."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsuperviseddeephomographyral2018_amd import _lib, ops  # noqa: E402
from tools.microbench import make_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='64,240,320,128,45')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--warm', type=int, default=0, help='1: no eviction before the traced launch (back-to-back, Infinity-Cache warm)')
    a = ap.parse_args()
    B, H, W, P, rho = (int(v) for v in a.shape.split(','))
    dev = torch.device('cuda:0')
    U, pts1, h4p, idx, I2 = make_inputs(B, H, W, P, rho, dev)
    _, theta = ops.solve_dlt(pts1, h4p, img_w=W, img_h=H); theta = theta.detach().contiguous()
    lib = _lib.load()
    dbg = C.CDLL(_lib.LIB_PATH)
    if not hasattr(dbg, 'uh_debug_set_trace'):
        raise SystemExit('this library was not built with -DUH_WARP_TRACE (set UH_LIB_PATH to the trace variant)')
    nblk = B * ((W + 63) // 64) * ((H + 15) // 16)
    nw = nblk * 4
    tr = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
    out = torch.empty_like(U)
    p = lambda t: C.c_void_p(t.data_ptr())
    ev_a = torch.empty((1 << 30) // 4, device=dev).normal_(); ev_b = torch.empty_like(ev_a)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    run = lambda: _lib.check(lib.uh_warp_forward(p(U), p(theta), p(out), None, B, H, W, 3, H, W, st), 'uh_warp_forward')
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    assert dbg.uh_debug_set_trace(p(tr)) == 0
    rows = []
    for rep in range(a.reps):
        tr.zero_()
        if not a.warm:
            ev_b.copy_(ev_a)                                   # the state the train step leaves the caches in
        torch.cuda.synchronize()
        _lib.profile_enable(True, only=('warp_forward',))
        run()
        torch.cuda.synchronize()
        pr = _lib.profile_read(); _lib.profile_enable(False)
        ev_us = pr['warp_forward'][0] / max(pr['warp_forward'][1], 1) * 1e3
        t = tr.cpu().numpy().reshape(nw, 16).astype(np.int64)
        widx = np.arange(nw)[(t[:, 8] > 0) & (t[:, 9] > 0)]     # wave record index = (virtual block id) * 4 + wave
        t = t[(t[:, 8] > 0) & (t[:, 9] > 0)]
        rt0, rt1 = t[:, 8], t[:, 9]                            # 100 MHz ticks
        origin = rt0.min()
        s_us = (rt0 - origin) / 100.0; e_us = (rt1 - origin) / 100.0
        span = float(e_us.max())
        life_mt = (t[:, 7] - t[:, 0]).astype(np.float64); life_rt = (rt1 - rt0).astype(np.float64)
        ok = life_rt > 50
        mt_per_us = float(np.median(life_mt[ok] / (life_rt[ok] / 100.0))) if ok.any() else float('nan')   # shader clock, MHz
        # resident waves over time on a 0.1 us grid
        grid = np.arange(0.0, span + 0.1, 0.1)
        occ = (np.searchsorted(np.sort(s_us), grid, side='right') - np.searchsorted(np.sort(e_us), grid, side='right')).astype(np.float64)
        plateau = float(np.percentile(occ[occ > 0], 90)) if (occ > 0).any() else 0.0
        hi = np.nonzero(occ >= 0.9 * plateau)[0]
        ramp = float(grid[hi[0]]) if hi.size else float('nan')
        tail = float(span - grid[hi[-1]]) if hi.size else float('nan')
        data_landed = s_us + (t[:, 3] - t[:, 0]) / mt_per_us    # per wave: when its data was there (DMA landed / loads returned)
        stores_issued = s_us + (t[:, 4] - t[:, 0]) / mt_per_us
        xcc = t[:, 10] & 0xF
        names = {0: 'A', 1: 'B', 2: 'C1', 3: 'C2'}
        okx = lambda x: (xcc == x) & ok
        per_xcd = {int(x): {'waves': int((xcc == x).sum()), 'first_start_us': round(float(s_us[xcc == x].min()), 2),
                            'last_end_us': round(float(e_us[xcc == x].max()), 2),
                            'end_us_all_but_64_waves': round(float(np.sort(e_us[xcc == x])[-65]), 2) if (xcc == x).sum() > 65 else None,
                            # is a slow XCD slow because of its WORK (share of gather waves) or because of its CLOCK (shader ticks per us)?
                            'gather_share': round(float(((t[:, 5] == 1) | (t[:, 5] == 3))[xcc == x].mean()), 3),
                            'shader_ticks_per_us': round(float(np.median(life_mt[okx(x)] / (life_rt[okx(x)] / 100.0))), 0) if okx(x).any() else None,
                            'mean_wave_life_us': round(float(life_rt[xcc == x].mean() / 100.0), 2),
                            'sum_of_wave_lifetimes_ms': round(float(life_rt[xcc == x].sum() / 100.0 / 1e3), 2)} for x in np.unique(xcc)}
        paths = {names[k]: {'share': round(float((t[:, 5] == k).mean()), 4),
                            'mean_life_us': round(float(life_rt[t[:, 5] == k].mean() / 100.0), 2),
                            'mean_start_us': round(float(s_us[t[:, 5] == k].mean()), 2)} for k in names if (t[:, 5] == k).any()}
        # who makes the tail: path mix of the waves that END in the last 2 us / the last 1 us, and of the last generation (the
        # waves that start after 90 % of all waves have started), with their lifetimes -- would dispatching gather tiles first help?
        late2, late1 = e_us > span - 2.0, e_us > span - 1.0
        lastgen = s_us >= np.percentile(s_us, 90)
        mix = lambda m: {names[k]: round(float((t[m, 5] == k).mean()), 3) for k in names if m.any() and (t[m, 5] == k).any()}
        life_by = lambda m: {names[k]: round(float(life_rt[m & (t[:, 5] == k)].mean() / 100.0), 2) for k in names if (m & (t[:, 5] == k)).any()}
        tail_who = {'ending_in_last_2us': {'waves': int(late2.sum()), 'mix': mix(late2)},
                    'ending_in_last_1us': {'waves': int(late1.sum()), 'mix': mix(late1)},
                    'last_generation_(last_10pct_to_start)': {'mix': mix(lastgen), 'mean_life_us_by_path': life_by(lastgen),
                                                              'p99_life_us': round(float(np.percentile(life_rt[lastgen], 99) / 100.0), 2),
                                                              'mean_end_us_by_path': {names[k]: round(float(e_us[lastgen & (t[:, 5] == k)].mean()), 2)
                                                                                      for k in names if (lastgen & (t[:, 5] == k)).any()}},
                    'all_waves_mix': mix(np.ones(len(t), bool))}
        # the stragglers themselves: the 24 waves that end last -- where they sit, when they started, which phase was long
        tiles_x, tiles_y = (W + 63) // 64, (H + 15) // 16
        order = np.argsort(-e_us)[:24]
        ph = lambda i, a_, b_: round(float((t[i, b_] - t[i, a_]) / mt_per_us), 2)
        stragglers = [{'end_us': round(float(e_us[i]), 2), 'start_us': round(float(s_us[i]), 2), 'life_us': round(float(life_rt[i] / 100.0), 2),
                       'path': names.get(int(t[i, 5]), '?'), 'xcd': int(xcc[i]), 'image': int(widx[i] // 4 // (tiles_x * tiles_y)),
                       'tile_y': int((widx[i] // 4 % (tiles_x * tiles_y)) // tiles_x), 'tile_x': int((widx[i] // 4 % (tiles_x * tiles_y)) % tiles_x),
                       'wave': int(widx[i] % 4), 'cu_hwid': int(t[i, 11] & 0xFFFF),
                       'coords_us': ph(i, 0, 1), 'issue_us': ph(i, 1, 2), 'wait_data_us': ph(i, 2, 3), 'consume_us': ph(i, 3, 4), 'store_drain_us': ph(i, 4, 7)}
                      for i in order]
        typical = {'coords_us': round(float(np.median((t[:, 1] - t[:, 0]) / mt_per_us)), 2), 'issue_us': round(float(np.median((t[:, 2] - t[:, 1]) / mt_per_us)), 2),
                   'wait_data_us': round(float(np.median((t[:, 3] - t[:, 2]) / mt_per_us)), 2), 'consume_us': round(float(np.median((t[:, 4] - t[:, 3]) / mt_per_us)), 2),
                   'store_drain_us': round(float(np.median((t[:, 7] - t[:, 4]) / mt_per_us)), 2),
                   'p99_wait_data_us': round(float(np.percentile((t[:, 3] - t[:, 2]) / mt_per_us, 99)), 2),
                   'p99_store_drain_us': round(float(np.percentile((t[:, 7] - t[:, 4]) / mt_per_us, 99)), 2),
                   'p999_store_drain_us': round(float(np.percentile((t[:, 7] - t[:, 4]) / mt_per_us, 99.9)), 2),
                   'p999_wait_data_us': round(float(np.percentile((t[:, 3] - t[:, 2]) / mt_per_us, 99.9)), 2)}
        ends_sorted = np.sort(e_us)
        row = {'rep': rep, 'shape': a.shape, 'tail_who': tail_who, 'stragglers': stragglers, 'median_and_p99_phases': typical,
               'end_us_when_all_but_N_waves_are_done': {str(n): round(float(ends_sorted[-n - 1]), 2) for n in (1000, 256, 64, 16, 4, 0)}, 'input': 'warm' if a.warm else 'cold (1 GiB evicting copy before the launch)',
               'event_duration_us': round(ev_us, 2), 'wave_span_us': round(span, 2),
               'dispatch_overhead_us': round(ev_us - span, 2), 'waves': int(len(t)), 'blocks': nblk,
               'shader_clock_MHz_during_launch': round(mt_per_us, 0),
               'mean_wave_life_us': round(float(life_rt.mean() / 100.0), 2), 'p99_wave_life_us': round(float(np.percentile(life_rt, 99) / 100.0), 2),
               'plateau_resident_waves': plateau, 'avg_resident_waves': round(float(life_rt.sum() / 100.0 / span), 1),
               'ramp_us_to_90pct_plateau': round(ramp, 2), 'tail_us_below_90pct_plateau': round(tail, 2),
               'first_wave_data_landed_us': round(float(data_landed.min()), 2), 'first_stores_issued_us': round(float(stores_issued.min()), 2),
               'last_wave_start_us': round(float(s_us.max()), 2),
               'waves_started_in_first_us': int((s_us < 1.0).sum()), 'waves_started_in_first_2us': int((s_us < 2.0).sum()),
               'per_xcd': per_xcd, 'paths': paths}
        rows.append(row)
        print(json.dumps(row), flush=True)
    med = lambda k: round(float(np.median([r[k] for r in rows])), 2)
    keys = ('event_duration_us', 'wave_span_us', 'dispatch_overhead_us', 'ramp_us_to_90pct_plateau', 'tail_us_below_90pct_plateau',
            'first_wave_data_landed_us', 'first_stores_issued_us', 'mean_wave_life_us', 'last_wave_start_us', 'avg_resident_waves')
    print(json.dumps({'summary_median_of_%d' % len(rows): {k: med(k) for k in keys}, 'shape': a.shape,
                      'input': rows[0]['input'] if rows else None}), flush=True)


if __name__ == '__main__':
    main()
