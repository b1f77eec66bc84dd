#!/bin/bash
# interleaved A/B of environment settings on the default bench: tools/ab.sh "VAR=1" "VAR=2 OTHER=1" ...  (three rounds)
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "round $round  [$cfg]  $r ms"
  done
done
