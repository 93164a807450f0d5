#!/bin/bash
# The whole GPU suite five times over on one box at the final binary (VERDICT r05 item 1 asked for consecutive green runs: flakiness, not a single pass,
# is what a driver run samples), plus smoke() once.  One summary line per run; the deterministic parity numbers of every run (-s) for comparison.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r06c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $OUT/repeat5.log
for i in 1 2 3 4 5; do
  timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "passed|failed|error|vs fp64 anchor|all networks|cutoff fixture" | sed "s/^/run $i: /" >> $OUT/repeat5.log
done
cat $OUT/repeat5.log
