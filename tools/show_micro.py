#!/usr/bin/env python
import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/micro_c.log'):
    d = json.loads(l)
    p = d['lib_prof_us']
    print('%-20s id=%d B%d %dx%d copy %6.1f | fwd %6.1f (%.3f) bwd %6.1f (%.3f) fused %5.1f | interleaved: fwd %6.1f bwd %6.1f fused %5.1f' % (
        d['lib'], d.get('identity', 0), d['B'], d['H'], d['W'], d['copy_us'], d['warp_fwd_us'], d['warp_fwd_frac'],
        d['warp_bwd_us'], d['warp_bwd_frac'], d['patch_fused_us'], p['warp_forward'], p['warp_backward'], p['warp_patch_l1_fused']))
