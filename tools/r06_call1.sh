#!/bin/bash
# Round-6 evidence call 1 (run through gpurun from the repo root): build() exercised for real on the GPU box, the InstanceNorm-backward
# traffic probe (alone / behind a dirty-L2 writer; durations, then FETCH_SIZE and WRITE_SIZE in separate PMC passes), the all-reduce probe
# as a ONE-rank RCCL self-run, the multi-step parity tests three times over.  Outputs under gpurun_out/r06/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
{
  echo "# build() on the GPU box: every header touched -> all 17 units recompiled by hipcc here, linked, loaded, smoke() against the oracle"
  date -u; hostname; /opt/rocm/bin/hipcc --version | head -2
  echo "shipped library (cross-compiled in the build container): $(sha256sum maskcyclegan-vc_amd/lib/libmcvc_hip.so)"
  ls -l --time-style=full-iso maskcyclegan-vc_amd/build/*.o | awk '{print $5, $6, $7, $9}'
  touch maskcyclegan-vc_amd/csrc/mcvc_common.h
  T0=$(date +%s)
  python __graft_entry__.py smoke
  echo "rc=$?  build + smoke: $(( $(date +%s) - T0 )) s"
  echo "rebuilt on the box: $(sha256sum maskcyclegan-vc_amd/lib/libmcvc_hip.so)"
  ls -l --time-style=full-iso maskcyclegan-vc_amd/build/*.o | awk '{print $5, $6, $7, $9}'
} > $OUT/build_on_box.log 2>&1

{
  for M in alone after_writer; do python tools/norm_bwd_probe.py $M; done
  for M in alone after_writer; do
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/nb_$M$C
      timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/nb_$M$C -o x -- python tools/norm_bwd_probe.py $M > /dev/null 2>&1
      echo "## $M $C (per dispatch; FETCH_SIZE x 2 per the guide's gfx950 correction, units KiB... see tools/pmc_traffic.py)"
      python tools/rocpd_pmc.py $(find /tmp/nb_$M$C -name "*.db" | head -1) norm_bwd
    done
  done
} > $OUT/norm_bwd_probe.log 2>&1

timeout 600 python tools/allreduce_probe.py --one-rank > $OUT/allreduce_probe_one_rank.json 2> $OUT/allreduce_probe_one_rank.err
timeout 600 python bench.py --rccl-one-rank --cpu-iters 0 > $OUT/bench_rccl_one_rank.json 2> $OUT/bench_rccl_one_rank.err

for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_hip_engine.py tests/test_hip_parity_fp64.py -q -x -m gpu -s 2>&1 | grep -v "^$" | tail -25
done > $OUT/parity_repeat3.log 2>&1

timeout 600 python bench.py > $OUT/bench_default_call1.json 2> $OUT/bench_default_call1.err
ls -la $OUT
