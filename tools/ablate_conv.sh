# timing ablations of conv_direct_kernel on one layer (results are wrong by construction)
for bits in 0 1 2 3 4 8 7 15; do
  echo -n "dbg=$bits  "; MCVC_CONV_DEBUG=$bits timeout 100 python tools/conv_microbench.py ${1:-up2} --batch ${2:-2} --iters 30 2>/dev/null | tail -1
done
