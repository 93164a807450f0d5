#!/bin/bash
# One batch size, quickly: the bench line (family table) and the rocprofv3 kernel-trace table of the serial schedule.
#   tools/quick_profile.sh <batch> [steps]        (through gpurun, from the repo root; outputs under gpurun_out/quick/)
B=${1:-32}; ST=${2:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd $R
mkdir -p gpurun_out/quick
python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps $ST --warmup 3 > gpurun_out/quick/bench_bs$B.json 2>/dev/null
rm -rf /tmp/p1; timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o x -- python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps $ST --warmup 4 --no-trace --serial > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/p1 -name "*.db" | head -1) $ST > gpurun_out/quick/kernel_stats_bs${B}_serial.txt 2>&1
python - <<PY
import json
r=json.loads(open('gpurun_out/quick/bench_bs$B.json').read().strip().splitlines()[-1])
print(r['ms_per_step'], json.dumps(r['kernel_time_ms_per_step']))
PY
head -40 gpurun_out/quick/kernel_stats_bs${B}_serial.txt
