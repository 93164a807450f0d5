mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 100 python bench.py --steps 10 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); k=d['kernel_time_ms_per_step']
print('%.2f ms/step  convM=%.2f convQ=%.2f convT=%.2f convL=%.2f wgrad25=%.2f norm=%.2f' % (d['ms_per_step'], k.get('conv_direct<2,1,2,2>',0), k.get('conv_direct<1,1,2,2>',0), k.get('conv_direct<2,1,4,1>',0), k.get('conv_direct<2,2,2,2>',0), k.get('conv_wgrad<2,5>',0), k.get('norm_fwd',0)+k.get('norm_bwd',0)))"; }
run MCVC_CONV_LDS_KB=78
run MCVC_CONV_LDS_KB=52
run MCVC_CONV_LDS_KB=39
run MCVC_CONV_LDS_KB=78 MCVC_CONV_Q_BELOW=512
run MCVC_CONV_LDS_KB=39 MCVC_CONV_Q_BELOW=512
run MCVC_CONV_LDS_KB=32 MCVC_CONV_Q_BELOW=100000
run MCVC_CONV_LDS_KB=130
