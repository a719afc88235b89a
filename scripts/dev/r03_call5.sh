#!/bin/bash
# GPU box, round 3 fifth call: full GPU suite, default bench line, 2-rank strong-scaling rehearsal over gloo (ranks share the GPU)
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03e; mkdir -p $O
if [ "$1" != "skiptests" ]; then timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { tail -40 $O/pytest_all.txt; exit 1; }; fi
tail -3 $O/pytest_all.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03e/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "scaling", "n_gpus")}, d["roofline"]["frac"], d["roofline"]["traffic"], d.get("fast_mode", {}).get("ms_per_step"), {k: v["ms_per_step"] for k, v in d.get("also", {}).items()}, d["parity_gate"]["index_mismatches"], d["cpu_baseline"]["value"])
PY
LIPVQ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --sustained 100 > $O/bench_gloo2_strong.json 2> $O/bench_gloo2.err || { tail -5 $O/bench_gloo2.err; exit 1; }
LIPVQ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --sustained 100 --scaling weak > $O/bench_gloo2_weak.json 2>> $O/bench_gloo2.err || { tail -5 $O/bench_gloo2.err; exit 1; }
python - <<'PY'
import json
for f in ("strong", "weak"):
    d = json.load(open(f"gpurun_out/r03e/bench_gloo2_{f}.json"))
    print(f, {k: d[k] for k in ("value", "ms_per_step", "scaling", "n_gpus")}, d["config"]["rows_per_gpu"], d["config"]["global_rows"], d["config"]["global_usage_rows_last_step"])
PY
