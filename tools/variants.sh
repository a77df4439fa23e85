#!/bin/bash
# Build A/B variants of the library (developer tool): tools/variants.sh name "-DFLAG ..." ...
set -e
cd "$(dirname "$0")/../unsuperviseddeephomographyral2018_amd/csrc"
OUT=../lib/variants; mkdir -p $OUT
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fvisibility=hidden $flags \
     uh_dlt.hip uh_warp.hip uh_misc.hip uh_patch.hip uh_losses.hip uh_inputs.hip uh_tail.hip uh_epilogue.hip -o $OUT/libuh_$name.so &
done
wait
ls -la $OUT
