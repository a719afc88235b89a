#!/bin/bash
# GPU box: rocprofv3 --kernel-trace of ONE bench process that runs as the headline does -- sustained loop first, warm-up, K timed
# steps -- and a summary restricted to the timed steps' dispatches (scripts/prof_summary.py).  Output under gpurun_out/<out>/:
#   kernel_stats_timed_<wl>.csv   per kernel: all dispatches vs the last K (the timed steps)
#   bench_profiled_<wl>.json      the bench line of THAT process (its ms_per_step is what the kernel averages must add up to)
# Usage: bash scripts/prof_stats.sh <outdir under gpurun_out> <workload> [steps, default 20] [bench args]
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; WL=$2; STEPS=${3:-20}; shift 2; [ $# -gt 0 ] && shift
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$WL -- python3 bench.py --workload $WL --steps $STEPS --warmup 5 --no-cpu-baseline --sustained 1000 --metric-only --traffic off "$@" > $OUT/bench_profiled_$WL.json 2> $OUT/prof_$WL.err
python3 scripts/prof_summary.py $OUT/trace_$WL $STEPS $OUT/kernel_stats_timed_$WL.csv
rm -rf $OUT/trace_$WL
head -6 $OUT/kernel_stats_timed_$WL.csv | cut -c1-160
python3 -c "import json,sys; d=json.load(open('$OUT/bench_profiled_$WL.json')); print('bench line of the profiled process: ms_per_step', d['ms_per_step'], 'ms_per_launch (hipEvents)', d['roofline']['ms_per_launch'])"
