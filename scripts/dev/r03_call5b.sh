#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { grep -E "^E|Error|FAILED" $O/pytest_all.txt | head -30; tail -5 $O/pytest_all.txt; exit 1; }
tail -3 $O/pytest_all.txt
bash scripts/dev/r03_call5.sh skiptests
