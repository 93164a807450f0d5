#!/bin/bash
# Same-box A/B of ONE planner knob on the bf16 inference forward (experiments build):
#   gpurun -- 'bash tools/ab_infer_knob.sh MCVC_BF16_S2_WIDE "0 1"'
L=$(pwd)/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
KNOB=$1; VALS=${2:-"0 1"}
for rep in 1 2 3; do for v in $VALS; do
  env MCVC_LIB=$L $KNOB=$v python bench.py --mode infer --dtype bf16 --steps 50 --warmup 10 --cpu-iters 0 --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('infer bf16 $KNOB=$v', round(r['ms_per_step'],4), 'parity', r.get('parity_vs_cpu_rel_l2'))"
done; done
