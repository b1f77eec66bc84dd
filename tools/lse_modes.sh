#!/bin/bash
# the two per-process modes of the LSE batch-4096 step: ms/step (and the host's enqueue time per step) of N separate processes
# tools/lse_modes.sh N "<bench flags>" ["ENV=.." ...]
cd "$(dirname "$0")/.."
N=${1:-8}; FLAGS=${2:---config=lse_small}; shift 2
for v in "${@:-X=0}"; do
  out=""
  for i in $(seq $N); do
    r=$(env $v python bench.py --steps 200 --warmup 20 --repeats 3 $FLAGS --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'enq', d['timing'].get('host_enqueue_ms_per_step'))")
    out="$out | $r"
  done
  echo "[$FLAGS] [$v]$out"
done
