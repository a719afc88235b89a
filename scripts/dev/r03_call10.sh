#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03j; mkdir -p $O
export LIPVQ_SCREEN_MONITOR=0
for m in fine coarse; do
for f in test_gpu_screen test_gpu_fused test_gpu_big_parity test_gpu_random_shapes; do
  LIPVQ_SCREEN_MODE=$m timeout -k 10 600 python -m pytest tests/$f.py -q -m gpu > $O/pytest_${m}_$f.txt 2>&1
  echo "$m $f: $(tail -1 $O/pytest_${m}_$f.txt)"; grep -E "^FAILED" $O/pytest_${m}_$f.txt | head -10
done; done
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 50 --warmup 20"
for wl in cfg2 cfg3 icrt; do
  for m in fine coarse; do
    LIPVQ_SCREEN_MODE=$m timeout -k 10 200 python bench.py --workload $wl $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $m ms_per_step %.4f launch(events, all kernels) %.4f exact_rows %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['rows_decided_by_exact_kernel']))" >> $O/coarse_ab.txt
  done
done
cat $O/coarse_ab.txt
export TMPDIR=/tmp
for wl in cfg2 cfg3 icrt; do
LIPVQ_SCREEN_MODE=coarse rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --sustained 0 --metric-only --traffic off > /dev/null 2> $O/prof_$wl.err
cp $(ls $O/trace_$wl/*/*kernel_stats.csv | head -1) $O/kernel_stats_coarse_$wl.csv; head -5 $O/kernel_stats_coarse_$wl.csv | cut -c1-70,200-260
rm -rf $O/trace_$wl
done
for wl in cfg2 cfg3; do for m in coarse; do
echo "== $wl $m" >> $O/stamps_coarse.txt
LIPVQ_SCREEN_MODE=$m LIPVQ_HIP_LIBRARY=build_ab/st_coarse/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py $wl 2>&1 | grep -v amdgpu | head -12 >> $O/stamps_coarse.txt
done; done
cat $O/stamps_coarse.txt
