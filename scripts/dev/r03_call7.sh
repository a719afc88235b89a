#!/bin/bash
# GPU box, round 3 seventh call: the one-product ("coarse") screen -- parity suites under LIPVQ_SCREEN_MODE=coarse, timings vs fine
cd "$(dirname "$0")/../.."
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_screen.py -x -q -m gpu -k "coarse_error_bound or error_bound" > $O/pytest_bound.txt 2>&1; tail -5 $O/pytest_bound.txt
for f in test_gpu_screen test_gpu_fused test_gpu_big_parity test_gpu_random_shapes test_gpu_module; do
  LIPVQ_SCREEN_MODE=coarse timeout -k 10 600 python -m pytest tests/$f.py -q -m gpu > $O/pytest_coarse_$f.txt 2>&1
  echo "$f: $(tail -1 $O/pytest_coarse_$f.txt)"; grep -E "^FAILED" $O/pytest_coarse_$f.txt | head -20
done
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 50 --warmup 20"
for rep in 1 2; do
for wl in cfg2 cfg3 icrt; do
  for m in fine coarse; do
    LIPVQ_SCREEN_MODE=$m timeout -k 10 200 python bench.py --workload $wl $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $m ms_per_step %.4f launch(events, both kernels) %.4f exact_rows %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['rows_decided_by_exact_kernel']))" >> $O/coarse_ab.txt
  done
done
done
cat $O/coarse_ab.txt
export TMPDIR=/tmp
for wl in cfg2 cfg3; do
LIPVQ_SCREEN_MODE=coarse rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --sustained 0 --metric-only --traffic off > /dev/null 2> $O/prof_$wl.err
cp $(ls $O/trace_$wl/*/*kernel_stats.csv | head -1) $O/kernel_stats_coarse_$wl.csv; head -4 $O/kernel_stats_coarse_$wl.csv | cut -c1-150
rm -rf $O/trace_$wl
done
