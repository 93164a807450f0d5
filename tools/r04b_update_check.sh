#!/bin/bash
# r4 (second session): fused optimizer step + re-pack -- parity tests, then same-box A/B against the two-launch form
mkdir -p gpurun_out
python -m pytest tests/test_hip_update.py -x -q 2>&1 | tail -15 > gpurun_out/upd_tests.log
python -m pytest tests/test_hip_engine.py tests/test_hip_twin.py tests/test_hip_cli.py -x -q 2>&1 | tail -15 > gpurun_out/upd_engine_tests.log
for rep in 1 2; do
for B in 1 8 32; do
  ST=40; if [ $B -ge 8 ]; then ST=12; fi
  for v in 1 0; do
    echo "== bs=$B MCVC_FUSED_UPDATE=$v"
    env MCVC_FUSED_UPDATE=$v python bench.py --batch-size $B --steps $ST --warmup 6 --cpu-iters 0 --no-extra-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernel_time_ms_per_step']; print(r['ms_per_step'], 'adam', k.get('adam'), 'pack', k.get('pack'), 'hbm', r.get('hbm_bytes_per_step'), 'launches', r.get('launches_per_step'))"
  done
done
done > gpurun_out/upd_ab.log 2>&1
