#!/bin/bash
# session AA: 2 pixels per lane (16x8 wave tiles) in the forward / in the backward, with smaller LDS slices and 8 waves per SIMD
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG="128,240,320,128,45;128,480,640,128,64;64,240,320,128,45"
for n in fs2l3o8 bs2l3; do
UH_LIB_PATH=$V/libuh_$n.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "warp or config4 or chain or literal or patch_backward" > gpurun_out/r2aa_pytest_$n.log 2>&1
tail -2 gpurun_out/r2aa_pytest_$n.log
done
: > gpurun_out/r2aa_micro.jsonl
timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2aa_micro.jsonl
for n in fs2 fs2l3o8 fs2l4 bs2 bs2l3; do
  UH_LIB_PATH=$V/libuh_$n.so timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2aa_micro.jsonl
done
timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2aa_micro.jsonl
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r2aa_micro.jsonl'):
    d=json.loads(l); print('%-18s B%3d %dx%d fwd %6.1f bwd %6.1f'%(d.get('lib','?'), d['B'], d['H'], d['W'], d['lib_prof_us'].get('warp_forward'), d['lib_prof_us'].get('warp_backward')))
PY
