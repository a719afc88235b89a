#!/bin/bash
# GPU box, round 3 fourth call: the (waves, row groups) shapes of tokenize_kernel -- parity suites and same-box timings
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03d; mkdir -p $O
for sh in w8rg2 w4rg2 w4rg1; do
  LIPVQ_TOK_SHAPE=$sh timeout -k 10 500 python -m pytest tests/test_gpu_big_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_fused.py tests/test_gpu_screen.py -x -q -m gpu > $O/pytest_$sh.txt 2>&1 || { tail -30 $O/pytest_$sh.txt; exit 1; }
  echo "$sh: $(tail -1 $O/pytest_$sh.txt)"
done
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 50 --warmup 20"
for rep in 1 2; do
for wl in cfg2 cfg3 icrt; do
  for sh in w8rg1 w8rg2 w4rg2 w4rg1; do
    LIPVQ_TOK_SHAPE=$sh timeout -k 10 200 python bench.py --workload $wl $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $sh ms_per_step %.4f launch %.4f exact_rows %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['rows_decided_by_exact_kernel']))" >> $O/shapes_ab.txt
  done
done
done
cat $O/shapes_ab.txt
