#!/bin/bash
# GPU box: the measurements a round's profiles/ entries are taken from.  Usage: bash scripts/dev/evidence.sh <tag, e.g. r02_d>
set -e
cd "$(dirname "$0")/../.."
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
for wl in cfg3 icrt; do python bench.py --workload $wl > $OUT/bench_$wl.json 2>> $OUT/bench.err; done
for wl in cfg2 cfg3 icrt; do bash scripts/prof_stats.sh $TAG $wl > /dev/null; done
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/train -- python3 $OLDPWD/scripts/profile_train_step_big.py > /dev/null 2>&1 )
cp $(ls $OUT/train/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_train_step_cfg2.csv
{ python tests/bench_train_step.py; python scripts/measure_small.py; python scripts/measure_wgrad.py; python scripts/dev/measure_mlp3_bwd.py; python scripts/dev/measure_mlp3_fwd.py;
  python scripts/dev/measure_scatter.py; python scripts/dev/measure_scatter_det.py; python scripts/dev/measure_stream.py; python scripts/measure_default.py; python scripts/dev/measure_embed_bwd.py; python scripts/dev/measure_vqvae.py; python scripts/dev/measure_bin_train.py; python scripts/dev/measure_mse.py; } > $OUT/side_measurements.txt 2>&1
rm -rf $OUT/train $OUT/trace_*
ls $OUT
