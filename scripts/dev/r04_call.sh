set -e
mkdir -p gpurun_out/r04g
python -m pytest tests -m gpu -x -q > gpurun_out/r04g/pytest.txt 2>&1 || { tail -40 gpurun_out/r04g/pytest.txt; exit 1; }
tail -2 gpurun_out/r04g/pytest.txt
for rep in 1 2; do
for wl in cfg2 cfg3 icrt; do
python build_ab/r03_tree/scripts/dev/shard_sweep.py $wl > gpurun_out/r04g/sweep_${wl}_r03_$rep.txt 2>&1
python scripts/dev/shard_sweep.py $wl > gpurun_out/r04g/sweep_${wl}_now_$rep.txt 2>&1
done; done
tail -n 6 gpurun_out/r04g/sweep_*.txt | grep -v "^$"
