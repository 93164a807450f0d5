#!/bin/bash
# The adopted tile order (gemm2: 8 row tiles per group where a point has >= 8 column tiles; igemm: 8 rows where the grid holds a full resident round
# per XCD) against the r5 order on the experiments build, same box, three alternations; then the operator / model parity files on the shipped library.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
L=$R/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
run() { local label=$1 B=$2 ST=$3; shift 3
  env MCVC_LIB=$L "$@" python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B %-10s %8.3f ms' % ('$label', r['ms_per_step']))"; }
{
for rep in 1 2 3; do
  for B in 1 8 32; do
    ST=40; if [ $B -ge 8 ]; then ST=12; fi; if [ $B -ge 32 ]; then ST=6; fi
    run r5order $B $ST MCVC_GEMM_MGROUP=0 MCVC_IGEMM_GROUP_ROWS=0
    run new $B $ST MCVC_GEMM_MGROUP=8
  done
done
} > $OUT/ab_tile_order_adopted.log 2>&1
cat $OUT/ab_tile_order_adopted.log
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_twin.py -q -m gpu -x 2>&1 | tail -4 > $OUT/ops_model_tests_tile_order.log
cat $OUT/ops_model_tests_tile_order.log
