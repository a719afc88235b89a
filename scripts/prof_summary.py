"""Summarise a rocprofv3 --kernel-trace run of bench.py so that the numbers are those of the bench line's TIMED steps:
    python scripts/prof_summary.py <trace dir> <steps> [out.csv]
For every kernel: dispatches in the whole process, mean / min duration over all of them, and mean / min over the LAST `steps`
dispatches -- the timed steps of bench.py (--steps K), which run behind its sustained loop and warm-up, i.e. in the clock state
the headline is measured in.  (rocprofv3's own *_kernel_stats.csv averages every dispatch of the process, cold ones included:
round 3's committed average, 480 us, was larger than the whole driver-timed step, 432 us.)"""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], int(sys.argv[2])
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = [("kernel", "dispatches", "mean_us_all", "min_us_all", f"mean_us_last_{steps}", f"min_us_last_{steps}", f"max_us_last_{steps}")]
for k, v in sorted(rows.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    v.sort()
    dur = [x[1] / 1e3 for x in v]
    last = dur[-steps:]
    out.append((k, len(dur), f"{sum(dur) / len(dur):.2f}", f"{min(dur):.2f}", f"{sum(last) / len(last):.2f}", f"{min(last):.2f}", f"{max(last):.2f}"))
w = csv.writer(open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout)
w.writerows(out)
