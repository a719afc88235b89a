set -e
mkdir -p gpurun_out/r04t
LIPVQ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 6 --warmup 2 --sustained 50 > gpurun_out/r04t/bench_gloo2_strong.json 2> gpurun_out/r04t/bench_gloo2.err || { tail -20 gpurun_out/r04t/bench_gloo2.err; exit 1; }
wc -l gpurun_out/r04t/bench_gloo2_strong.json
python -c "
import json; d=json.load(open('gpurun_out/r04t/bench_gloo2_strong.json')); print(d['value'], d['n_gpus'], d['scaling'], d['config']['rows_per_gpu'], d['config']['global_usage_rows_last_step'], d['config']['parallelism'])"
python scripts/dev/measure_train_big.py > gpurun_out/r04t/train_big2.txt 2>&1 || true; tail -3 gpurun_out/r04t/train_big2.txt
python tests/bench_train_step.py > gpurun_out/r04t/train_small.txt 2>&1 || true; tail -4 gpurun_out/r04t/train_small.txt
