#!/bin/bash
# ms per step at the per-rank shapes of the 8-GPU metric and the other configs: tools/r03_shapes.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-base}
OUT=gpurun_out/shapes_$TAG.txt
: > $OUT
run() {
  r=$(python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-extra-legs --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], round(d['value']/1e6,2))")
  echo "$* : $r" | tee -a $OUT
}
for b in 6400 12800 25600; do run --batch $b; done
STEPS=100 run
STEPS=100 run --update-method full_adam
STEPS=50 run --config large_tables
run --config lse_small
