#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03zz; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -30 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 && tail -2 $O/smoke.txt
bash scripts/dev/evidence.sh r03zz > $O/evidence.log 2>&1 || { tail -20 $O/evidence.log; exit 1; }
cat $O/bench.json | cut -c1-400
