#!/bin/bash
# Round-6 evidence in ONE gpurun call at revision fc7f430: tools/profile_round.sh (PMC traffic, SQ passes, bench lines, traces, kernel tables, timelines) for
# bs = 1 / 8 / 32 + the bf16 inference forward, then the whole GPU test suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/profile_round.sh r06 fc7f430 "1 8 32" infer > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $R/gpurun_out/r06/gpu_tests.log
ls -la $R/gpurun_out/r06 | wc -l
