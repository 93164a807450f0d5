#!/bin/bash
# Same-box sweep of ONE planner knob of the experiments build:  bash tools/knob_sweep.sh MCVC_SGEMM_WGS "256 128 64" 8
E=$(pwd)/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
B=${3:-1}; ST=40; if [ $B -ge 8 ]; then ST=12; fi; if [ $B -ge 32 ]; then ST=6; fi
for rep in 1 2; do for v in $2; do
  env MCVC_LIB=$E $1=$v python bench.py --batch-size $B --steps $ST --warmup 5 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B $1=$v', round(r['ms_per_step'],3), 'launches', r.get('kernel_launches_per_step'), {k: round(x,2) for k,x in list(r.get('kernel_time_ms_per_step',{}).items())[:6]})"
done; done
