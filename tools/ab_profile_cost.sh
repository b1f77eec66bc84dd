for r in 1 2; do for sh in "--config=lse_small" "--batch=6400" "--batch=51200"; do for f in "--no-profile" ""; do
  v=$(python bench.py --steps 200 --warmup 20 --repeats 3 $sh --no-cpu-baseline --no-extra-legs $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "[$sh] [${f:-profile loss+gather}] $v"
done; done; done
