#!/bin/bash
# Same-box A/B of the tile ORDER of the two GEMM families on the experiments build: row tiles per group of gemm2_kernel (MCVC_GEMM_MGROUP: 0 = the
# column tile fastest = shipped) and of igemm_kernel (MCVC_IGEMM_GROUP_KB / _MAX: 2048 KB of A panels = shipped; 65536 + MAX m = m row tiles).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp; cd $R
L=$R/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
run() {   # run <label> <batch> <steps> env...
  local label=$1 B=$2 ST=$3; shift 3
  env MCVC_LIB=$L "$@" python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernel_time_ms_per_step']; print('bs=$B %-28s %8.3f ms  wino_gemm %7.3f  sgemm %7.3f' % ('$label', r['ms_per_step'], k.get('wino_gemm',0), k.get('sgemm',0)))"
}
{
for rep in 1 2; do
  for B in 32 8; do
    ST=6; if [ $B -le 8 ]; then ST=12; fi
    run base $B $ST MCVC_GEMM_MGROUP=0
    run gemm_mg4 $B $ST MCVC_GEMM_MGROUP=4
    run gemm_mg8 $B $ST MCVC_GEMM_MGROUP=8
    run igemm_mg8 $B $ST MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=8
    run igemm_mg4 $B $ST MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=4
    run both_mg8 $B $ST MCVC_GEMM_MGROUP=8 MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=8
  done
done
run base 1 30 MCVC_GEMM_MGROUP=0
run both_mg8 1 30 MCVC_GEMM_MGROUP=8 MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=8
run base 1 30 MCVC_GEMM_MGROUP=0
run both_mg8 1 30 MCVC_GEMM_MGROUP=8 MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=8
} > gpurun_out/r06b/ab_tile_order.log 2>&1
cat gpurun_out/r06b/ab_tile_order.log
