#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 500 python scripts/dev/soak.py 300 2>&1 | grep -v amdgpu > $O/soak.txt; tail -4 $O/soak.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], d['fast_mode']['ms_per_step'], {k:(v['ms_per_step'], v['frac']) for k,v in d['also'].items()}, d['parity_gate']['index_mismatches'], d['cpu_baseline']['value'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
