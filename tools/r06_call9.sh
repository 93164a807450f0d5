#!/bin/bash
# Round-6 call 9: slab sums with independent loads (norm_*_reg, act_*_vec, mask_grad): parity + same-box A/B (ab_a = before, ab_b = after)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_twin.py -q -x -m gpu 2>&1 | tail -4 > $OUT/slab_fix_tests.log
timeout 900 python -m pytest tests/test_hip_engine.py -q -x -m gpu -k "bit_reproducible or deferred or golden or unmodified or cutoff" 2>&1 | tail -4 >> $OUT/slab_fix_tests.log
bash tools/ab_lib.sh "1 8 32" > $OUT/ab_slab_loads.log 2>&1
ls -la $OUT | tail -3
