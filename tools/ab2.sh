#!/bin/bash
# as tools/ab.sh with extra bench flags in $BENCH_FLAGS
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --no-profile $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "round $round  [$cfg]  $r ms"
  done
done
