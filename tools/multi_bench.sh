cd "$(dirname "$0")/.."
for m in full_adam dense_adam adagrad sgd; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --update-method $m > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --uniform-words > gpurun_out/bench_uniform.json 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --host-batches > gpurun_out/bench_hostbatches.json 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --read-cost-every 1 > gpurun_out/bench_readcost.json 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --config large_tables > gpurun_out/bench_large.json 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --config lse_small > gpurun_out/bench_small.json 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['update_method'], d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-300:])
PY
