#!/bin/bash
# A/B of the F(4x4,5x5) path (MCVC_WINO4_NB = samples per pass from which upSample1/2 take it; 0 = off): the GPU tests, then step times.
# Run through gpurun from the repo root.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in "1 0" "1 1" "1 2" "2 0" "2 2" "2 4" "4 0" "4 4" "8 0" "8 4"; do
  set -- $cfg
  echo "== bs=$1 MCVC_WINO4_NB=$2"
  MCVC_WINO4_NB=$2 python bench.py --batch-size $1 --steps 30 --warmup 5 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print(r['ms_per_step'])"
done
