#!/bin/bash
# Round-6 call 20: is the 0.3 ms of the 50-step bf16 bench in the working tree a ONE-TIME event early in the process?  warm-up 10 / 40 / 100 x 50 timed steps, both trees
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{ for W in 10 40 100; do for t in _old . _old .; do cd $R/$t
  python bench.py --mode infer --dtype bf16 --steps 50 --warmup $W --cpu-iters 0 --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('warmup $W tree=$t', round(r['ms_per_step'],4))"
done; done; } > $OUT/c2d1d_warmup_sweep.log 2>&1
cat $OUT/c2d1d_warmup_sweep.log
