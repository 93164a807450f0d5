#!/bin/bash
# Round-6 call 14: conv2dto1d + InstanceNorm of the bf16 forward in one launch (parity, same-box A/B, kernel table)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 python -m pytest tests/test_hip_bf16.py -q -m gpu 2>&1 | tail -12 > $OUT/bf16_tests_c2d1d.log
timeout 900 bash tools/ab_infer_knob.sh MCVC_BF16_C2D1D_FUSED "0 1" > $OUT/ab_c2d1d_fused.log 2>&1
rm -rf /tmp/prof_i; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_i -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_i -name "*.db" | head -1) 10 > $OUT/kernel_stats_infer_bf16_c2d1d.txt 2>&1
timeout 600 python bench.py --mode infer --dtype bf16 > $OUT/bench_infer_bf16_c2d1d.json 2>/dev/null
ls -la $OUT | tail -3
