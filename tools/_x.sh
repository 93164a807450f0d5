export TMPDIR=/tmp
rm -rf /tmp/pmc_x; timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_BUSY_CYCLES -d /tmp/pmc_x -o x -- python bench.py --mode infer --dtype bf16 --cpu-iters 0 --steps 10 --warmup 3 --no-trace > /dev/null 2>&1
echo rc=$?
DB=$(find /tmp/pmc_x -name "*.db" | head -1)
mkdir -p gpurun_out; python tools/rocpd_pmc.py $DB > gpurun_out/x_pmc.txt
python tools/rocpd_stats.py $DB 10 | head -7 | cut -c1-150
