#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { grep -E "^E|FAILED" $O/pytest_all.txt | head -20; }
tail -2 $O/pytest_all.txt
export LIPVQ_SCREEN_MONITOR=0
LIPVQ_SCREEN_MODE=coarse timeout -k 10 600 python -m pytest tests/test_gpu_screen.py tests/test_gpu_random_shapes.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python scripts/dev/soak.py 100 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 600 python scripts/dev/coarse_sweep.py 2>&1 | grep -v amdgpu > $O/coarse_sweep.txt; cat $O/coarse_sweep.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3 -- python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --sustained 0 --metric-only --traffic off > /dev/null 2> $O/prof_cfg3.err
cp $(ls $O/trace_cfg3/*/*kernel_stats.csv | head -1) $O/kernel_stats_cfg3.csv; head -5 $O/kernel_stats_cfg3.csv | cut -c1-60,190-260
rm -rf $O/trace_cfg3
