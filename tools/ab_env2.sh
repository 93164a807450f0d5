#!/bin/bash
# A/B of environment combinations over batch sizes:  tools/ab_env2.sh "<batch sizes>" "VAR=a VAR2=b" "VAR=c" ...   (through gpurun, from the repo root)
BATCHES=$1; shift
for B in $BATCHES; do
  ST=30; if [ $B -ge 8 ]; then ST=10; fi
  for combo in "$@"; do
    echo "== bs=$B $combo"
    env $combo python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernel_time_ms_per_step']; print(r['ms_per_step'], 'launches', r['kernel_launches_per_step'])"
  done
done
