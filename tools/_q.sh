cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4
python bench.py --no-extra-configs --batch-size 32 --cpu-iters 0 --steps 6 --warmup 3 > gpurun_out/w4/bench_bs32.json 2>/dev/null
rm -rf /tmp/p1; timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o x -- python bench.py --no-extra-configs --batch-size 32 --cpu-iters 0 --steps 6 --warmup 4 --no-trace --serial > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/p1 -name "*.db" | head -1) 6 > gpurun_out/w4/kernel_stats_bs32_serial.txt 2>&1
head -45 gpurun_out/w4/kernel_stats_bs32_serial.txt
