"""Dev tool: per-step kernel time breakdown from a rocprofv3 kernel_stats.csv: python scripts/kernel_breakdown.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / steps / 1e3:.1f} us per step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f'{r["Name"][:64]:64s} calls/step {int(r["Calls"]) / steps:5.1f}  avg {float(r["AverageNs"]) / 1e3:7.1f} us  per step {float(r["TotalDurationNs"]) / steps / 1e3:7.1f} us')
