#!/bin/bash
# Round-6 call 13: bisect which hunk of the slab-sum restructure moves the deterministic fp64-anchor numbers (var_v1: norm_fwd_reg, v2: norm_bwd_reg,
# v3: act_*_vec, v4: mask_grad; each on top of the old kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
L=$R/maskcyclegan-vc_amd/lib
for v in v1 v2 v3 v4; do
  echo "== lib var_$v"; MCVC_LIB=$L/var_$v.so timeout 900 python -m pytest tests/test_hip_parity_fp64.py -q -m gpu -s 2>&1 | grep -E "mode vs fp64|all networks|passed|failed"
done > $OUT/fp64_bisect_slab.log 2>&1
cat $OUT/fp64_bisect_slab.log
