#!/bin/bash
# Same-box A/B of the loss readback lag (bench.py --loss-lag 1 | 2):   gpurun -- 'bash tools/ab_loss_lag.sh "1 8 32"'
for B in ${1:-1}; do
  ST=50; if [ $B -ge 8 ]; then ST=16; fi; if [ $B -ge 32 ]; then ST=8; fi
  for rep in 1 2 3; do for v in 1 2; do
    python bench.py --batch-size $B --steps $ST --warmup 6 --cpu-iters 0 --no-extra-configs --no-trace --loss-lag $v 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B lag=$v', round(r['ms_per_step'],3))"
  done; done
done
