cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in a b; do
  rm -rf /tmp/prof_$v
  MCVC_LIB=$R/maskcyclegan-vc_amd/lib/ab_$v.so timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$v -o x -- python bench.py --no-extra-configs --batch-size $1 --cpu-iters 0 --steps 6 --warmup 3 --no-trace --serial > /dev/null 2>&1
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB 6 > gpurun_out/ab_stats_bs$1_$v.txt 2>&1
done
