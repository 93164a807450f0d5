#!/bin/bash
# Round-6 call 7: fused residual-block layers of the bf16 forward (parity, same-box A/B, kernel table), then the whole GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu 2>&1 | tail -15 > $OUT/bf16_tests_trunk.log
bash tools/ab_infer_knob.sh MCVC_BF16_TRUNK_FUSED "0 1" > $OUT/ab_trunk_fused.log 2>&1
rm -rf /tmp/prof_i; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_i -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_i -name "*.db" | head -1) 10 > $OUT/kernel_stats_infer_bf16_trunk.txt 2>&1
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $OUT/gpu_tests_mid.log
ls -la $OUT
