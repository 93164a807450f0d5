#!/bin/bash
# Round-6 call 12: which change moved the deterministic fp64-anchor numbers?  The same test with the slab-sum restructure (ab_b) and without (ab_a).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
L=$R/maskcyclegan-vc_amd/lib
for v in a b; do
  echo "== lib ab_$v"; MCVC_LIB=$L/ab_$v.so timeout 900 python -m pytest tests/test_hip_parity_fp64.py -q -m gpu -s 2>&1 | grep -E "mode vs fp64|all networks|passed|failed|pooled"
done > $OUT/fp64_ab_slab.log 2>&1
ls -la $OUT | tail -2
