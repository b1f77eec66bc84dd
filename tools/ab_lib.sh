#!/bin/bash
# interleaved A/B of library builds: tools/ab_lib.sh libA.so libB.so ...   (paths relative to cunvsm_amd/)
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "$@"; do
    r=$(CUNVSM_AMD_LIB=$PWD/cunvsm_amd/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-profile $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "round $round  [$lib]  $r ms"
  done
done
