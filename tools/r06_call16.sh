#!/bin/bash
# Round-6 call 16: is the bf16 forward slower / noisier with bf16_c2d1d_kernel in the translation unit?  _old/ = the committed tree (no such
# kernel), . = the working tree; production libraries, alternating runs on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3 4; do
  for t in _old .; do
    cd $R/$t
    python bench.py --mode infer --dtype bf16 --steps 50 --warmup 10 --cpu-iters 0 --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('infer bf16 tree=$t', round(r['ms_per_step'],4))"
  done
done > $OUT/ab_tree_c2d1d.log 2>&1
cat $OUT/ab_tree_c2d1d.log
