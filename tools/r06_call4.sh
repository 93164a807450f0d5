#!/bin/bash
# Round-6 call 4: fused last conv of the bf16 forward (parity, same-box A/B, kernel table), then the bs=1 kernel tables (serial / concurrent)
# after the InstanceNorm-backward fix.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_bf16.py -q -x -m gpu 2>&1 | tail -8 > $OUT/bf16_tests_last.log
bash tools/ab_infer_knob.sh MCVC_BF16_LAST_FUSED "0 1" > $OUT/ab_last_fused.log 2>&1
rm -rf /tmp/prof_i; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_i -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_i -name "*.db" | head -1) 10 > $OUT/kernel_stats_infer_bf16_last.txt 2>&1
for MODE in serial concurrent; do
  F=""; if [ $MODE = serial ]; then F="--serial"; fi
  rm -rf /tmp/prof_$MODE; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$MODE -o x -- python bench.py --no-extra-configs --batch-size 1 --cpu-iters 0 --steps 30 --warmup 4 --no-trace $F > /dev/null 2>&1
  DB=$(find /tmp/prof_$MODE -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB 30 > $OUT/mid_kernel_stats_bs1_$MODE.txt 2>&1
  python tools/rocpd_timeline.py $DB > $OUT/mid_timeline_bs1_$MODE.txt 2>&1
done
ls -la $OUT
