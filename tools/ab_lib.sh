#!/bin/bash
# Same-box A/B of two builds of libmcvc_hip.so (box-to-box spread is +-2 %, more than most single changes):
#   cp maskcyclegan-vc_amd/lib/libmcvc_hip.so maskcyclegan-vc_amd/lib/ab_a.so      (build A, e.g. `git stash` + build)
#   ... change, rebuild ...   cp .../libmcvc_hip.so .../ab_b.so
#   gpurun -- 'bash tools/ab_lib.sh "1 2 8"'            (alternates A, B three times per batch size)
L=$(pwd)/maskcyclegan-vc_amd/lib
for B in ${1:-1}; do
  ST=30; if [ $B -ge 8 ]; then ST=10; fi
  for rep in 1 2 3; do for v in a b; do
    MCVC_LIB=$L/ab_$v.so python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B $v', round(r['ms_per_step'],3))"
  done; done
done
