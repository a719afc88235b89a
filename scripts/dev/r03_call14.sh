#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03n; mkdir -p $O
export LIPVQ_SCREEN_MONITOR=0 LIPVQ_SCREEN_MODE=coarse
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 30 --warmup 10"
for v in ca_base ca_nostage ca_nobar ca_noldsb ca_notrack ca_base; do
  echo "== $v (one-product screen)" >> $O/cfg3_coarse_ablation.txt
  LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload cfg3 $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 ms_per_step %.4f launch %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))" >> $O/cfg3_coarse_ablation.txt
done
cat $O/cfg3_coarse_ablation.txt
