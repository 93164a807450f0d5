#!/bin/bash
# Same-box A/B of ONE planner knob on the experiments build (MCVC_EXPERIMENTS=1 python __graft_entry__.py -> lib/libmcvc_hip_exp.so):
#   gpurun -- 'bash tools/ab_knob.sh MCVC_WGEMM "0 1" "1 8 32"'      (alternates the values three times per batch size)
L=$(pwd)/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
KNOB=$1; VALS=${2:-"0 1"}
for B in ${3:-1}; do
  ST=30; if [ $B -ge 8 ]; then ST=10; fi; if [ $B -ge 32 ]; then ST=6; fi
  for rep in 1 2 3; do for v in $VALS; do
    env MCVC_LIB=$L $KNOB=$v python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B $KNOB=$v', round(r['ms_per_step'],3))"
  done; done
done
