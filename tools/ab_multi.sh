#!/bin/bash
# Same-box comparison of several builds:  bash tools/ab_multi.sh "a b c" "1 8 32"   (libs lib/ab_<v>.so; three alternating rounds per batch size)
L=$(pwd)/maskcyclegan-vc_amd/lib
for B in ${2:-1}; do
  ST=40; if [ $B -ge 8 ]; then ST=12; fi; if [ $B -ge 32 ]; then ST=6; fi
  for rep in 1 2 3; do for v in $1; do
    MCVC_LIB=$L/ab_$v.so python bench.py --batch-size $B --steps $ST --warmup 5 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B $v', round(r['ms_per_step'],3), r['roofline']['kernel'], round(r['roofline']['avg_launch_ms']*1e3,2), 'us', r.get('kernel_time_ms_per_step',{}).get('wino_gemm'))"
  done; done
done
