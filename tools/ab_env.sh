#!/bin/bash
# A/B of an environment knob over batch sizes:  tools/ab_env.sh <VAR> "<values>" "<batch sizes>"   (through gpurun, from the repo root)
VAR=$1; VALS=$2; BATCHES=${3:-"32"}
for B in $BATCHES; do
  ST=30; if [ $B -ge 8 ]; then ST=10; fi
  for v in $VALS; do
    echo "== bs=$B $VAR=$v"
    env $VAR=$v python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernel_time_ms_per_step']; print(r['ms_per_step'], 'wino_gemm', k.get('wino_gemm'), 'sgemm', k.get('sgemm'))"
  done
done
