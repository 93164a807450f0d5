#!/bin/bash
# Round-6 call 3: the InstanceNorm backward without its scratch copy of the argument struct -- parity, same-box A/B (lib/ab_a.so = before,
# lib/ab_b.so = after) at bs = 1 / 8 / 32, the isolated probe again (durations + the two PMC passes).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -q -x -m gpu 2>&1 | tail -5 > $OUT/norm_fix_tests.log
bash tools/ab_lib.sh "1 8 32" > $OUT/ab_norm_bwd_scratch.log 2>&1
{
  python tools/norm_bwd_probe.py alone
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/nb_$C; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/nb_$C -o x -- python tools/norm_bwd_probe.py alone > /dev/null 2>&1
    echo "## alone $C (KiB per dispatch)"; python tools/rocpd_pmc.py $(find /tmp/nb_$C -name "*.db" | head -1) norm_bwd
  done
} > $OUT/norm_bwd_probe_after.log 2>&1
ls -la $OUT
