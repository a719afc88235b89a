#!/bin/bash
# GPU box: boxes of this pool grant the same kernel different shader clocks (profiles/r04_i_clock_ab.txt).  On a box that runs
# this tree's cfg2 launch >100 MHz below the round-3 tree's, measure the variants given; elsewhere stop after the probe.
#   bash scripts/dev/clock_bisect.sh <outdir under gpurun_out> <variant lib names under build_ab ...>
cd "$(dirname "$0")/../.."
OUT=gpurun_out/$1; shift; mkdir -p $OUT
python scripts/dev/clock_ab.py cfg2 1 build_ab/r03_tree . 2>/dev/null > $OUT/probe.txt
cat $OUT/probe.txt | tail -2
old=$(awk '$2=="build_ab/r03_tree"{print $5}' $OUT/probe.txt); new=$(awk '$2=="."{print $5}' $OUT/probe.txt)
if [ $((old - new)) -lt 100 ]; then echo "box grants both trees about the same clock ($old / $new MHz): nothing to bisect here"; exit 0; fi
args="build_ab/r03_tree ."
for v in "$@"; do args="$args .::build_ab/$v/_lipvq_hip.so"; done
python scripts/dev/clock_ab.py cfg2 2 $args 2>/dev/null > $OUT/bisect_cfg2.txt
cat $OUT/bisect_cfg2.txt
