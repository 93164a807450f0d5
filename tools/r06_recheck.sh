#!/bin/bash
# Re-check after the build container was re-created (every .o / .so rebuilt from source by build()): smoke, the whole GPU suite, one default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06b/smoke.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r06b/gpu_tests.log
timeout 600 python bench.py > gpurun_out/r06b/bench_default.json 2> gpurun_out/r06b/bench_default.err
tail -3 gpurun_out/r06b/gpu_tests.log; tail -2 gpurun_out/r06b/smoke.log; python -c "
import json; d=json.load(open('gpurun_out/r06b/bench_default.json')); print(d['ms_per_step'], d['value'], [c.get('ms_per_step') for c in d.get('configs',[])], d.get('after_identity_cutoff',{}).get('ms_per_step'))"
