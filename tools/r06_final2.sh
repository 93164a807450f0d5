#!/bin/bash
# Round-6 evidence, second edition, in ONE gpurun call at revision 3f1edef (the tile orders of gemm2 / igemm / wgemm changed after fc7f430): tools/profile_round.sh
# (PMC traffic, SQ passes, bench lines, traces, kernel tables, timelines) for bs = 1 / 8 / 32 + the bf16 inference forward, then the whole GPU test suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/profile_round.sh r06 3f1edef "1 8 32" infer > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $R/gpurun_out/r06/gpu_tests.log
tail -3 $R/gpurun_out/r06/gpu_tests.log
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_default.json')); print(d['ms_per_step'], d['value'], [c.get('ms_per_step') for c in d.get('configs',[])], d.get('after_identity_cutoff',{}).get('ms_per_step'), d['hbm_bytes_ratio_to_algorithmic'])"
