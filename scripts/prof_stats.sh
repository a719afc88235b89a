#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of one bench workload; copies the kernel_stats csv + the bench line to gpurun_out/<out>/
# Usage: bash scripts/prof_stats.sh <outdir under gpurun_out> <workload> [bench args]
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; WL=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -- python3 bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --sustained 0 --metric-only "$@" > $OUT/bench_profiled_$WL.json 2> $OUT/prof_$WL.err
f=$(ls $OUT/trace_$WL/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats_$WL.csv
head -8 $OUT/kernel_stats_$WL.csv | cut -c1-200
