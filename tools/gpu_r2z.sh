#!/bin/bash
# session Z: backward register diet (UH_WARP_BWD_DIET=1: 77-80 VGPRs -> 6 waves/SIMD instead of 5): parity tests with the
# variant library, then A/B/A/B microbench at the north_star point and at config 4
mkdir -p gpurun_out; cd /root/repo
V=unsuperviseddeephomographyral2018_amd/lib/variants
CFG="128,240,320,128,45;128,480,640,128,64"
UH_LIB_PATH=$V/libuh_diet.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -q --tb=short -x -k "backward or chain or gradient or patch or dU" > gpurun_out/r2z_pytest.log 2>&1
tail -3 gpurun_out/r2z_pytest.log
: > gpurun_out/r2z_micro.jsonl
for rep in 1 2; do
  timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2z_micro.jsonl
  UH_LIB_PATH=$V/libuh_diet.so timeout 120 python tools/microbench.py --iters 40 --configs "$CFG" 2>/dev/null >> gpurun_out/r2z_micro.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r2z_micro.jsonl'):
    d=json.loads(l); print(d.get('lib','?'), d['H'], d['W'], 'fwd', d['lib_prof_us'].get('warp_forward'), 'bwd', d['lib_prof_us'].get('warp_backward'), d.get('warp_bwd_frac'))
PY
