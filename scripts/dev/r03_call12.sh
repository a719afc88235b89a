#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03l; mkdir -p $O
export LIPVQ_SCREEN_MONITOR=0
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 50 --warmup 20"
for rep in 1 2; do for wl in cfg2 icrt; do for m in fine coarse; do
    LIPVQ_SCREEN_MODE=$m timeout -k 10 200 python bench.py --workload $wl $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $m ms_per_step %.4f launch(events, all kernels) %.4f exact_rows %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['rows_decided_by_exact_kernel']))" >> $O/coarse_ab.txt
done; done; done
cat $O/coarse_ab.txt
for m in fine coarse; do
echo "== cfg2 $m" >> $O/stamps.txt
LIPVQ_SCREEN_MODE=$m LIPVQ_HIP_LIBRARY=build_ab/st_coarse/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py cfg2 2>&1 | grep -v amdgpu >> $O/stamps.txt
done
cat $O/stamps.txt
