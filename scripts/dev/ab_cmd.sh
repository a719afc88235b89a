cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in cfg2 cfg3 icrt; do bash scripts/prof_stats.sh r02b $wl > gpurun_out/r02b_$wl.txt 2>&1; done
bash scripts/pmc_sq.sh r02b_sq > gpurun_out/r02b_sq.txt 2>&1
bash scripts/pmc_hbm.sh r02b_hbm cfg2 > gpurun_out/r02b_hbm.txt 2>&1
python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r02b_bench_cfg3.json 2>/dev/null
python bench.py --workload icrt --no-cpu-baseline > gpurun_out/r02b_bench_icrt.json 2>/dev/null
python tests/bench_train_step.py > gpurun_out/r02b_train.txt 2>&1
python scripts/measure_default.py > gpurun_out/r02b_default.txt 2>&1
python scripts/measure_wgrad.py > gpurun_out/r02b_wgrad.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02b_train -- python3 scripts/profile_train_step_big.py > gpurun_out/r02b_trainprof.log 2>&1
tail -c 400 gpurun_out/r02b_bench.json
