set -e
mkdir -p gpurun_out/r04r
python -m pytest tests -m gpu -x -q > gpurun_out/r04r/pytest.txt 2>&1 || { tail -40 gpurun_out/r04r/pytest.txt; exit 1; }
tail -2 gpurun_out/r04r/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()"
