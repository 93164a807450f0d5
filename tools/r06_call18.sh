#!/bin/bash
# Round-6 call 18: why is the bf16 forward 0.35 ms slower with bf16_c2d1d_kernel although that kernel is 42 us faster than what it replaces?
# old tree / new tree (production) / new tree experiments build with the knob off and on; shader clock sampled beside each run.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # label, dir, env...
  lab=$1; dir=$2; shift 2
  cd $R/$dir
  ( while true; do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.1; done ) > /tmp/clk_$lab.txt 2>/dev/null &
  CP=$!
  env "$@" python bench.py --mode infer --dtype bf16 --steps 600 --warmup 20 --cpu-iters 0 --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('$lab', round(r['ms_per_step'],4), 'host enqueue', r.get('host',{}))"
  kill $CP 2>/dev/null; wait $CP 2>/dev/null
  echo "   sclk samples: $(grep -o '([0-9]*Mhz)' /tmp/clk_$lab.txt | sort | uniq -c | sort -rn | head -4 | tr '\n' ' ')"
}
{
L=$R/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
for rep in 1 2; do
  run old_$rep _old X=1
  run newprod_$rep . X=1
  run newexp_k0_$rep . MCVC_LIB=$L MCVC_BF16_C2D1D_FUSED=0
  run newexp_k1_$rep . MCVC_LIB=$L MCVC_BF16_C2D1D_FUSED=1
done
} > $OUT/c2d1d_mystery.log 2>&1
cat $OUT/c2d1d_mystery.log
