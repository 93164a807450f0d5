#!/bin/bash
# bias gradients on the auxiliary stream behind their layer's weight gradient (MCVC_BIAS_AUX=1, candidate) against on the main stream in front of
# the data gradient (0 = shipped so far): experiments build, three alternations at bs = 1 / 8 / 32 (+ post-cut-off bs=1); then the model / engine /
# twin parity files on the shipped library (which has the candidate as its default).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r06c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
L=$R/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
run() { local label=$1 B=$2 ST=$3; shift 3
  env MCVC_LIB=$L "$@" python bench.py --batch-size $B --steps $ST --warmup 6 --cpu-iters 0 --no-extra-configs --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B %-10s %8.3f ms' % ('$label', r['ms_per_step']))"; }
{
for rep in 1 2 3 4; do
  for B in 1 8 32; do
    ST=60; if [ $B -ge 8 ]; then ST=16; fi; if [ $B -ge 32 ]; then ST=6; fi
    run main $B $ST MCVC_BIAS_AUX=0
    run aux $B $ST MCVC_BIAS_AUX=1
  done
done
} > $OUT/ab_bias_aux.log 2>&1
cat $OUT/ab_bias_aux.log
timeout 1500 python -m pytest tests/test_hip_model.py tests/test_hip_engine.py tests/test_hip_twin.py tests/test_hip_parity_fp64.py -q -m gpu -x 2>&1 | tail -4 > $OUT/tests_bias_aux.log
cat $OUT/tests_bias_aux.log
