#!/bin/bash
# GPU box: the measurements a round's profiles/ entries are taken from.  Usage: bash scripts/dev/evidence.sh <tag, e.g. r04_z> [r03 tree for the same-box sweep]
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/ with the tag as prefix.
set -e
cd "$(dirname "$0")/../.."
TAG=$1; OUT=gpurun_out/$TAG; OLD=${2:-}; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
for wl in cfg3 icrt; do python bench.py --workload $wl > $OUT/bench_$wl.json 2>> $OUT/bench.err; done
# warm-clock kernel traces of ONE bench process per workload, timed steps only (scripts/prof_stats.sh) + SQ counters of the same build
for wl in cfg2 cfg3 icrt; do bash scripts/prof_stats.sh $TAG $wl 20 > $OUT/prof_stats_$wl.txt 2>&1; done
for wl in cfg2 cfg3 icrt; do bash scripts/pmc_sq.sh $TAG/sq_$wl --workload $wl > /dev/null 2>&1 && cp $OUT/sq_$wl/sq_counters.txt $OUT/sq_counters_$wl.txt; rm -rf $OUT/sq_$wl; done
# shard sweep (the compute-side ceiling of strong scaling), optionally against an older tree on the same box
for wl in cfg2 cfg3 icrt; do
  [ -n "$OLD" ] && python $OLD/scripts/dev/shard_sweep.py $wl 2>&1 | grep -E "^ +[0-9]" | sed "s/^/$wl old /" >> $OUT/shard_sweep.txt
  python scripts/dev/shard_sweep.py $wl 2>&1 | grep -E "^ +[0-9]" | sed "s/^/$wl now /" >> $OUT/shard_sweep.txt
done
# training step: kernel stats + side measurements
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/train -- python3 $OLDPWD/scripts/profile_train_step_big.py > /dev/null 2>&1 )
cp $(ls $OUT/train/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_train_step_cfg2.csv; rm -rf $OUT/train
{ python tests/bench_train_step.py; python scripts/dev/measure_train_big.py; python scripts/measure_small.py; python scripts/dev/measure_vqvae.py; python scripts/measure_default.py; } > $OUT/side_measurements.txt 2>&1
ls $OUT
