#!/bin/bash
# Same-box A/B of two builds (lib/ab_a.so, lib/ab_b.so) on the bf16 inference forward (BASELINE configs[4]); see tools/ab_lib.sh
L=$(pwd)/maskcyclegan-vc_amd/lib
for rep in 1 2 3; do for v in a b; do
  MCVC_LIB=$L/ab_$v.so python bench.py --mode infer --dtype bf16 --steps 50 --warmup 10 --cpu-iters 0 --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('infer bf16 $v', round(r['ms_per_step'],4))"
done; done
