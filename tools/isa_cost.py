#!/usr/bin/env python
"""Static VALU-cycle estimate of a gfx950 kernel from hipcc -S output, using the issue costs measured by
tools/ubench (2 cycles: f32 add/sub/mul/fma, add_u32, and, mov; 8: rcp; ~5: v_pk_*; 4: the rest).
usage: isa_cost.py file.s substring [substring...]"""
import collections, re, sys
s = open(sys.argv[1]).read()
COST2 = {'v_add_f32', 'v_sub_f32', 'v_mul_f32', 'v_fma_f32', 'v_fmac_f32', 'v_add_u32', 'v_and_b32', 'v_mov_b32',
         'v_subrev_f32', 'v_mac_f32', 'v_sub_u32', 'v_or_b32', 'v_xor_b32', 'v_subrev_u32', 'v_fmaak_f32', 'v_fmamk_f32'}
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not all(t in name for t in sys.argv[2:]):
        continue
    ins = []
    for l in body.split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in ';.':
            continue
        ins.append(t.split()[0])
    h = collections.Counter(ins)
    cyc = 0
    for k, v in h.items():
        if not k.startswith('v_'):
            continue
        base = re.sub(r'_e(32|64)$', '', k)
        cyc += v * (2 if base in COST2 else 8 if 'rcp' in base else 5 if base.startswith('v_pk_') else 4)
    nv = sum(v for k, v in h.items() if k.startswith('v_'))
    print('%s\n  instr %d  valu %d  est. VALU cycles %d  vmem %d  salu %d' % (
        name, len(ins), nv, cyc, sum(v for k, v in h.items() if k.startswith(('buffer_', 'global_', 'flat_'))),
        sum(v for k, v in h.items() if k.startswith('s_'))))
    print('  ' + ' '.join('%s:%d' % kv for kv in sorted(h.items(), key=lambda kv: -kv[1])[:40]))
