#!/bin/bash
# Round-6 call 2: fused conv1 of the bf16 forward (parity, same-box A/B against prep + generic tile, kernel table), the FETCH_SIZE / WRITE_SIZE
# calibration probe (tools/pmc_calib.hip), the NaN-poisoned-allocator test.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_bf16.py -q -x -m gpu 2>&1 | tail -8 > $OUT/bf16_tests.log
bash tools/ab_infer_knob.sh MCVC_BF16_CONV1_FUSED "0 1" > $OUT/ab_conv1_fused.log 2>&1
rm -rf /tmp/prof_i; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_i -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_i -name "*.db" | head -1) 10 > $OUT/kernel_stats_infer_bf16_conv1.txt 2>&1
{
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/pmc_calib.hip -o /tmp/pmc_calib && /tmp/pmc_calib
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal_$C; timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/cal_$C -o x -- /tmp/pmc_calib > /dev/null 2>&1
    echo "## $C (raw counter value per dispatch; every kernel moves 268.4 MB in and / or 268.4 MB out)"
    python tools/rocpd_pmc.py $(find /tmp/cal_$C -name "*.db" | head -1) calib_
  done
} > $OUT/pmc_calib.log 2>&1
timeout 900 python -m pytest tests/test_hip_engine.py -q -x -m gpu -k "did_not_write or bit_reproducible" 2>&1 | tail -8 > $OUT/poison_test.log
ls -la $OUT
