set -e
mkdir -p gpurun_out/r04l
python -m pytest tests -m gpu -x -q > gpurun_out/r04l/pytest.txt 2>&1 || { tail -40 gpurun_out/r04l/pytest.txt; exit 1; }
tail -2 gpurun_out/r04l/pytest.txt
for rep in 1 2; do
for wl in cfg2 cfg3 icrt; do
python build_ab/r03_tree/scripts/dev/shard_sweep.py $wl 2>&1 | grep -E "^ +(1|8) " | sed "s/^/$wl r03 /" >> gpurun_out/r04l/ab.txt
SWEEP_G=1,8 python scripts/dev/shard_sweep.py $wl 2>&1 | grep -E "^ +[0-9]" | sed "s/^/$wl now /" >> gpurun_out/r04l/ab.txt
done; done
sort gpurun_out/r04l/ab.txt
python scripts/dev/rccl_contention.py cfg2 1000 > gpurun_out/r04l/rccl_contention_cfg2.txt 2>&1; cat gpurun_out/r04l/rccl_contention_cfg2.txt
python scripts/dev/rccl_contention.py cfg2 1000 65536 > gpurun_out/r04l/rccl_contention_cfg2_65536.txt 2>&1; cat gpurun_out/r04l/rccl_contention_cfg2_65536.txt
