#!/bin/bash
# A/B of weight-gradient planner knobs (run on the GPU box):  tools/ab_wgrad.sh "<cfg1>;<cfg2>;..." shapes...
IFS=';' read -ra CFGS <<< "$1"; shift
python tools/wgrad_microbench.py up2 --iters 50 > /dev/null 2>&1   # clocks / page-in
for cfg in "${CFGS[@]}"; do
  echo "== $cfg"
  env $cfg python tools/wgrad_microbench.py "$@" 2>&1 | grep -v "Warn\|amdgpu.ids"
done
