#!/bin/bash
# HBM-counter A/B of the GEMM tile orders (experiments build): FETCH_SIZE / WRITE_SIZE passes of the serial schedule, per kernel instance.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
export MCVC_LIB=$R/maskcyclegan-vc_amd/lib/libmcvc_hip_exp.so
pmc() { name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  rm -rf /tmp/pmc_$name; timeout 900 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o x -- "$@" > /dev/null 2>&1
  find /tmp/pmc_$name -name "*.db" | head -1; }
one() {  # label batch env...
  local label=$1 B=$2; shift 2
  CMD="env $* python bench.py --no-extra-configs --batch-size $B --cpu-iters 0 --steps 3 --warmup 2 --no-trace --serial"
  DBF=$(pmc f FETCH_SIZE -- $CMD); DBW=$(pmc w WRITE_SIZE -- $CMD)
  python tools/pmc_traffic.py $DBF $DBW 5 > $OUT/pmc_order_${label}_bs$B.json 2> $OUT/pmc_order_${label}_bs$B.err
  python - <<PY
import json
d=json.load(open("$OUT/pmc_order_${label}_bs$B.json"))
print("bs=$B %-10s total %.2f GB/step" % ("$label", d["hbm_bytes_per_step_pmc"]/1e9))
for k,v in d["instances"].items():
    if "gemm" in k and v["hbm_bytes_per_step"] > 3e8:
        print("    %-36s %3d launches  read %8.1f MB/launch  write %7.1f MB/launch" % (k, v["launches_per_step"], v["hbm_read_bytes_per_step"]/v["launches_per_step"]/1e6, v["hbm_write_bytes_per_step"]/v["launches_per_step"]/1e6))
PY
}
{
for B in 32 8; do
  one base $B MCVC_GEMM_MGROUP=0
  one mg8 $B MCVC_GEMM_MGROUP=8 MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=8
  one mg4 $B MCVC_GEMM_MGROUP=4 MCVC_IGEMM_GROUP_KB=65536 MCVC_IGEMM_GROUP_MAX=4
done
} > $OUT/pmc_order.log 2>&1
cat $OUT/pmc_order.log
