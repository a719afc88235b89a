#!/bin/bash
# GPU box: the kernels of ONE training step at the metric's batch, in launch order with durations and the gaps between them
# (rocprofv3 --kernel-trace of scripts/profile_train_step_big.py, last of its six steps).  Usage: bash scripts/dev/train_step_sequence.sh <outdir under gpurun_out>
set -e
cd "$(dirname "$0")/../.."
OUT=gpurun_out/$1; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/trace -- python3 $OLDPWD/scripts/profile_train_step_big.py > /dev/null 2>&1 )
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
# the last step: from the last tokenize_kernel dispatch back to the fills in front of it, forward to the end
ti = max(i for i, r in enumerate(rows) if "tokenize_kernel" in r[2])
prev = max(i for i, r in enumerate(rows[:ti]) if "tokenize_kernel" in r[2])
step = rows[prev:ti]                      # one whole step (tokenize of step k-1 ... just before tokenize of step k)
t0 = step[0][0]
with open(out + "/train_step_sequence.txt", "w") as fo:
    fo.write(f"# one training step, launch order: start us (from the step's tokenize launch), duration us, gap to the previous kernel's end us, kernel\n")
    last_end = step[0][0]
    tot = gaps = 0.0
    for s, e, k in step:
        fo.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - last_end) / 1e3:7.1f}  {k}\n")
        tot += (e - s) / 1e3; gaps += max(0, s - last_end) / 1e3; last_end = e
    fo.write(f"# {len(step)} launches, {tot:.1f} us of kernels, {gaps:.1f} us of gaps, {(step[-1][1] - t0) / 1e3:.1f} us first start -> last end\n")
print(open(out + "/train_step_sequence.txt").read())
PY
rm -rf $OUT/trace
