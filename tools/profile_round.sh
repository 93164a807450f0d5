#!/bin/bash
# Collect the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> <git_rev> "<batch sizes>"
# For every batch size: bench line (native), rocprofv3 kernel-trace stats of the serial and concurrent schedules, kernels-in-flight
# timeline, and -- in two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share one on gfx950) -- HBM bytes per launch and
# kernel family.  Outputs under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -u
TAG=${1:-r02p}; REV=${2:-unknown}; BATCHES=${3:-"1"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for B in $BATCHES; do
  ST=30; if [ $B -ge 8 ]; then ST=12; fi; if [ $B -ge 32 ]; then ST=6; fi
  CI=""; if [ $B -ge 32 ]; then CI="--cpu-iters 1"; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C; timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o x -- python bench.py --batch-size $B --cpu-iters 0 --steps 3 --warmup 2 --no-trace --serial > /dev/null 2>&1
  done
  python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) > $OUT/pmc_traffic_bs$B.json.tmp 2> $OUT/pmc_bs$B.err \
    && python -c "
import json,sys
j=json.load(open('$OUT/pmc_traffic_bs$B.json.tmp')); j['git_rev']='$REV'; j['batch_size']=$B
json.dump(j, open('$OUT/pmc_traffic_bs$B.json','w'), indent=1)" && rm -f $OUT/pmc_traffic_bs$B.json.tmp
  # the bench line quotes the traffic of THIS binary: refresh the file it reads before running it
  cp $OUT/pmc_traffic_bs$B.json $R/profiles/r02_pmc_traffic_bs$B.json 2>/dev/null
  timeout 900 python bench.py --batch-size $B $CI > $OUT/bench_bs$B.json 2> $OUT/bench_bs$B.err
  timeout 600 python bench.py --batch-size $B --cpu-iters 0 --steps $ST --warmup 4 --dump-trace $OUT/per_launch_trace_bs$B.txt > /dev/null 2>&1
  for MODE in serial concurrent; do
    F=""; if [ $MODE = serial ]; then F="--serial"; fi
    rm -rf /tmp/prof_$MODE; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$MODE -o x -- python bench.py --batch-size $B --cpu-iters 0 --steps $ST --warmup 4 --no-trace $F > /dev/null 2>&1
    DB=$(find /tmp/prof_$MODE -name "*.db" | head -1)
    python tools/rocpd_stats.py $DB $ST > $OUT/kernel_stats_bs${B}_$MODE.txt 2>&1
    python tools/rocpd_timeline.py $DB > $OUT/timeline_bs${B}_$MODE.txt 2>&1
  done
done
ls -la $OUT
