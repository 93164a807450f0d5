#!/bin/bash
# Round-6 call 10: slab sums with independent loads, proper A/B (ab_a = old norm / misc kernels, everything else current); the last conv's second form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
bash tools/ab_lib.sh "1 8 32" > $OUT/ab_slab_loads.log 2>&1
timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu 2>&1 | tail -4 > $OUT/bf16_tests_last2.log
bash tools/ab_infer_knob.sh MCVC_BF16_LAST_FUSED "0 1" > $OUT/ab_last_fused2.log 2>&1
rm -rf /tmp/prof_i; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_i -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_i -name "*.db" | head -1) 10 > $OUT/kernel_stats_infer_bf16_last2.txt 2>&1
ls -la $OUT | tail -4
