#!/bin/bash
# Round-6 call 17: kernels-in-flight timeline of the bf16 forward, working tree vs the committed tree (_old/): where do 0.35 ms per forward go?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in . _old; do
  cd $R/$t; n=$(echo $t | tr -d './'); n=${n:-new}
  rm -rf /tmp/p_$n; timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$n -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 --no-trace > /dev/null 2>&1
  python $R/tools/rocpd_timeline.py $(find /tmp/p_$n -name "*.db" | head -1) > $OUT/timeline_infer_$n.txt 2>&1
done
ls -la $OUT | tail -3
