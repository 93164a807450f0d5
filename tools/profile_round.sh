#!/bin/bash
# Collect a round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <round, e.g. r03> <git_rev> "<batch sizes>" [infer]
# For every batch size: two separate PMC passes for HBM bytes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950), two SQ passes
# (MFMA busy / instruction counts / wave occupancy / stall shares; LDS conflicts), the bench line (native), a per-launch trace, rocprofv3
# kernel-trace stats of the serial and the shipped schedule and the kernels-in-flight timeline.  Outputs under gpurun_out/<round>/ ;
# copy what is to be judged into profiles/.   (rocprofv3 runs: --kernel-trace only, or --kernel-trace + --pmc; never with other traces.)
set -u
RND=${1:-r03}; REV=${2:-unknown}; BATCHES=${3:-"1"}; INFER=${4:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$RND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
SQB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU"
pmc() { # name, counters..., then -- command
  name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  rm -rf /tmp/pmc_$name; timeout 900 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o x -- "$@" > /dev/null 2>&1
  find /tmp/pmc_$name -name "*.db" | head -1
}
for B in $BATCHES; do
  ST=30; if [ $B -ge 8 ]; then ST=12; fi; if [ $B -ge 32 ]; then ST=6; fi
  CI=""; if [ $B -ge 32 ]; then CI="--cpu-iters 1"; fi
  # (3 + 2 = 5 identical iterations: tools/pmc_traffic.py divides the dispatches behind the initial re-pack by that)
  CMD="python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps 3 --warmup 2 --no-trace --serial"
  DBF=$(pmc f FETCH_SIZE -- $CMD); DBW=$(pmc w WRITE_SIZE -- $CMD)
  python tools/pmc_traffic.py $DBF $DBW 5 > $OUT/pmc_traffic_bs$B.json.tmp 2> $OUT/pmc_bs$B.err \
    && python -c "
import json
j=json.load(open('$OUT/pmc_traffic_bs$B.json.tmp')); j['git_rev']='$REV'; j['batch_size']=$B
json.dump(j, open('$OUT/pmc_traffic_bs$B.json','w'), indent=1)" && rm -f $OUT/pmc_traffic_bs$B.json.tmp
  # the bench line quotes the traffic of THIS binary: refresh the file it reads before running it
  cp $OUT/pmc_traffic_bs$B.json $R/profiles/${RND}_pmc_traffic_bs$B.json 2>/dev/null
  DBA=$(pmc a $SQA -- $CMD); DBB=$(pmc b $SQB -- $CMD)
  { echo "# SQ counters per kernel (averages per dispatch), bs=$B, serial schedule, rev $REV"; echo "# pass A: $SQA"; python tools/rocpd_pmc.py $DBA; echo "# pass B: $SQB"; python tools/rocpd_pmc.py $DBB; } > $OUT/sq_bs$B.txt 2>&1
  python tools/sq_summary.py $DBA $DBB > $OUT/sq_summary_bs$B.txt 2>&1
  if [ $B -eq 1 ]; then timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; fi
  timeout 900 python bench.py --no-extra-configs --batch-size $B $CI > $OUT/bench_bs$B.json 2> $OUT/bench_bs$B.err
  timeout 600 python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps $ST --warmup 4 --dump-trace $OUT/per_launch_trace_bs$B.txt > /dev/null 2>&1
  for MODE in serial concurrent; do
    F=""; if [ $MODE = serial ]; then F="--serial"; fi
    rm -rf /tmp/prof_$MODE; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$MODE -o x -- python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps $ST --warmup 4 --no-trace $F > /dev/null 2>&1
    DB=$(find /tmp/prof_$MODE -name "*.db" | head -1)
    python tools/rocpd_stats.py $DB $ST > $OUT/kernel_stats_bs${B}_$MODE.txt 2>&1
    python tools/rocpd_timeline.py $DB > $OUT/timeline_bs${B}_$MODE.txt 2>&1
  done
  if [ $B -le 4 ]; then python tools/task_timeline.py --batch-size $B --steps 3 > $OUT/task_timeline_bs$B.txt 2>&1; fi
done
if [ -n "$INFER" ]; then
  ICMD="python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3"
  DBA=$(pmc ia $SQA -- $ICMD); DBB=$(pmc ib $SQB -- $ICMD)
  python tools/sq_summary.py $DBA $DBB > $OUT/sq_summary_infer_bf16.txt 2>&1
  timeout 600 python bench.py --mode infer --dtype bf16 > $OUT/bench_infer_bf16.json 2> $OUT/bench_infer_bf16.err
  timeout 600 python bench.py --mode infer --dtype f32 > $OUT/bench_infer_f32.json 2> $OUT/bench_infer_f32.err
  rm -rf /tmp/prof_i; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_i -o x -- $ICMD > /dev/null 2>&1
  python tools/rocpd_stats.py $(find /tmp/prof_i -name "*.db" | head -1) 10 > $OUT/kernel_stats_infer_bf16.txt 2>&1
fi
ls -la $OUT
