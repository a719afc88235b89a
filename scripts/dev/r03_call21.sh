#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { grep -E "^E|FAILED" $O/pytest_all.txt | head -20; }
tail -2 $O/pytest_all.txt
timeout -k 10 300 python scripts/dev/measure_vqvae.py 2>&1 | grep -v amdgpu > $O/vqvae.txt; cat $O/vqvae.txt
