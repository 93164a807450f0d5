#!/bin/bash
# Round-6 call 5: what the two-level accumulation of the batched Winograd GEMMs costs or gains in TIME (lib/ab_a.so = no fold, ab_b.so = fold at 32):
# every product of a bs=1 iteration alone on the chip (warm and cold operands), then the step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
L=$R/maskcyclegan-vc_amd/lib
{
for v in a b a b; do echo "== lib ab_$v warm"; MCVC_LIB=$L/ab_$v.so python tools/gemm_bs1_shapes.py 2>/dev/null | tail -25; done
for v in a b; do echo "== lib ab_$v cold"; MCVC_LIB=$L/ab_$v.so GEMM_COLD=1 python tools/gemm_bs1_shapes.py 2>/dev/null | tail -25; done
} > $OUT/ab_fold_gemm_shapes.log 2>&1
bash tools/ab_lib.sh "1 8" > $OUT/ab_fold_step.log 2>&1
ls -la $OUT
