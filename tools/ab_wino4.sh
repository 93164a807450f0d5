#!/bin/bash
# A/B of the F(4x4,.) paths: MCVC_WINO4_NB / MCVC_WINO43_NB = samples per pass from which upSample1/2 take F(4x4,5x5) / downSample1/2 take
# F(4x4,3x3); 0 = off.  Arguments: "bs nb4 nb43" triples.  Run through gpurun from the repo root.
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  echo "== bs=$1 MCVC_WINO4_NB=$2 MCVC_WINO43_NB=$3"
  ST=30; if [ $1 -ge 8 ]; then ST=10; fi
  MCVC_WINO4_NB=$2 MCVC_WINO43_NB=$3 python bench.py --batch-size $1 --steps $ST --warmup 5 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print(r['ms_per_step'])"
done
