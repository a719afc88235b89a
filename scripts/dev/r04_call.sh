set -e
mkdir -p gpurun_out/r04z3
python -m pytest tests -m gpu -x -q > gpurun_out/r04z3/pytest.txt 2>&1 || { tail -40 gpurun_out/r04z3/pytest.txt; exit 1; }
tail -2 gpurun_out/r04z3/pytest.txt
python scripts/dev/soak.py 60 > gpurun_out/r04z3/soak.txt 2>&1 || { tail -20 gpurun_out/r04z3/soak.txt; exit 1; }
tail -3 gpurun_out/r04z3/soak.txt
