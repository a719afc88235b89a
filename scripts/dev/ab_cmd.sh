cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in cfg2 cfg3 icrt; do bash scripts/prof_stats.sh r02c $wl > gpurun_out/r02c_$wl.txt 2>&1; done
bash scripts/pmc_sq.sh r02c_sq > gpurun_out/r02c_sq.txt 2>&1
bash scripts/pmc_hbm.sh r02c_hbm cfg2 > gpurun_out/r02c_hbm.txt 2>&1
python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r02c_bench_cfg3.json 2>/dev/null
python bench.py --workload icrt --no-cpu-baseline > gpurun_out/r02c_bench_icrt.json 2>/dev/null
python tests/bench_train_step.py > gpurun_out/r02c_train.txt 2>&1
python scripts/measure_default.py > gpurun_out/r02c_default.txt 2>&1
python scripts/measure_wgrad.py > gpurun_out/r02c_wgrad.txt 2>&1
python scripts/dev/measure_scatter.py > gpurun_out/r02c_scatter.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c_train -- python3 scripts/profile_train_step_big.py > gpurun_out/r02c_trainprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c_graph -- python3 scripts/dev/prof_graph_step.py > gpurun_out/r02c_graphprof.log 2>&1
tail -c 300 gpurun_out/r02c_bench.json
