#!/bin/bash
# wgemm_kernel: one contiguous range of tile ids per XCD (MCVC_WGEMM_XCD=1, adopted) against lid = blockIdx.x (0) on the experiments build:
# step time three times alternated at bs = 1 / 8 / 32, then the HBM counters per kernel instance at bs = 32 and 8.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
export MCVC_LIB=$R/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
run() { local label=$1 B=$2 ST=$3; shift 3
  env "$@" python bench.py --batch-size $B --steps $ST --warmup 4 --cpu-iters 0 --no-extra-configs --no-trace 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); print('bs=$B %-10s %8.3f ms' % ('$label', r['ms_per_step']))"; }
pmc() { name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  rm -rf /tmp/pmc_$name; timeout 900 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o x -- "$@" > /dev/null 2>&1
  find /tmp/pmc_$name -name "*.db" | head -1; }
one() { local label=$1 B=$2; shift 2
  CMD="env $* python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps 3 --warmup 2 --no-trace --serial"
  DBF=$(pmc f FETCH_SIZE -- $CMD); DBW=$(pmc w WRITE_SIZE -- $CMD)
  python tools/pmc_traffic.py $DBF $DBW 5 > $OUT/pmc_wgemm_${label}_bs$B.json 2> $OUT/pmc_wgemm_${label}_bs$B.err
  python - <<PY
import json
d=json.load(open("$OUT/pmc_wgemm_${label}_bs$B.json"))
print("bs=$B %-10s total %.2f GB/step" % ("$label", d["hbm_bytes_per_step_pmc"]/1e9))
for k,v in d["instances"].items():
    if "wgemm" in k:
        print("    %-36s %3d launches  read %8.1f MB/launch  write %7.1f MB/launch" % (k, v["launches_per_step"], v["hbm_read_bytes_per_step"]/v["launches_per_step"]/1e6, v["hbm_write_bytes_per_step"]/v["launches_per_step"]/1e6))
PY
}
{
for rep in 1 2 3; do
  for B in 1 8 32; do
    ST=40; if [ $B -ge 8 ]; then ST=12; fi; if [ $B -ge 32 ]; then ST=6; fi
    run blockidx $B $ST MCVC_WGEMM_XCD=0
    run xcd $B $ST MCVC_WGEMM_XCD=1
  done
done
for B in 32 8 1; do
  one blockidx $B MCVC_WGEMM_XCD=0
  one xcd $B MCVC_WGEMM_XCD=1
done
} > $OUT/ab_wgemm_xcd.log 2>&1
cat $OUT/ab_wgemm_xcd.log
