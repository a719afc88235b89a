"""Dev tool: the numbers of a scripts/dev/evidence.sh collection on one screen.   python scripts/dev/evidence_summary.py gpurun_out/<tag>"""
import json
import sys
d0 = sys.argv[1]
for wl, f in (("cfg2", "bench.json"), ("cfg3", "bench_cfg3.json"), ("icrt", "bench_icrt.json")):
    d = json.load(open(f"{d0}/{f}"))
    r = d["roofline"]
    print(wl, "value %.4g ms_per_step %.4f sustained %.4f" % (d["value"], d["ms_per_step"], d["sustained"]["ms_per_step"]))
    print("   ", {k: (round(r[k], 4) if isinstance(r.get(k), float) else r.get(k)) for k in (
        "frac", "frac_algorithmic_floor", "ms_per_launch", "mfma_busy_frac", "shader_clock_mhz", "frac_at_measured_clock",
        "frac_algorithmic_floor_at_measured_clock", "traffic")})
    print("    power", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in (r.get("power") or {}).items() if k != "how"})
    t = d.get("tuned")
    print("    tuned", t if isinstance(t, str) else (t["choice"], {k: round(v, 4) for k, v in t.get("ms_per_launch", {}).items()}))
    if "also" in d:
        print("    also", {k: (round(v["ms_per_step"], 3), round(v["frac"], 3)) for k, v in d["also"].items()})
    if "parity_gate" in d:
        g = d["parity_gate"]
        print("    gate", g["index_mismatches"], g["rows_compared"], g["max_rel_distance_gap_of_mismatches"])
    if "fast_mode" in d:
        print("    fast", round(d["fast_mode"]["ms_per_step"], 4), d["fast_mode"]["index_flip_rate_vs_parity"])
    if "cpu_baseline" in d:
        print("    cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"], "gpu/cpu", round(d["gpu_vs_cpu"]))
    print("    timed-step kernels:", [l.strip()[:100] for l in open(f"{d0}/kernel_stats_timed_{wl}.csv").read().splitlines()[1:4]])
    print("   ", open(f"{d0}/prof_stats_{wl}.txt").read().splitlines()[-1])
print(open(f"{d0}/shard_sweep.txt").read() and "".join(sorted(open(f"{d0}/shard_sweep.txt").readlines())))
print("".join(l for l in open(f"{d0}/side_measurements.txt") if "train step" in l or "VQVAE" in l))
