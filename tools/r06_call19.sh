#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{ for t in _old . _old .; do cd $R/$t; echo "== tree $t"; python $R/tools/infer_step_series.py 100 2>&1 | grep -v amdgpu.ids | cut -c1-1800; done; } > $OUT/infer_step_series.log 2>&1
cat $OUT/infer_step_series.log
