#!/bin/bash
# The two multi-step parity files after their live-reference gates were made host-independent (numbers printed with -s)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1200 python -m pytest tests/test_hip_parity_fp64.py tests/test_hip_engine.py -q -m gpu -s -k "accurate or cutoff or golden or reference_train or lr_decay" 2>&1 | grep -v "Warning\|detach\|return float" | tail -60 > gpurun_out/r06b/parity_host_independent.log
tail -5 gpurun_out/r06b/parity_host_independent.log
