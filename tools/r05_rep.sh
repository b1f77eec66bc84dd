#!/bin/bash
# distribution of ms/step over separate processes: tools/r05_rep.sh N "<bench flags>" ["ENV=.. ENV=.."]
cd "$(dirname "$0")/.."
N=$1; FLAGS=$2; shift 2
for v in "${@:-X=0}"; do
  out=""
  for i in $(seq $N); do
    r=$(env $v python bench.py --steps 100 --warmup 20 --repeats 3 $FLAGS --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    out="$out $r"
  done
  echo "[$FLAGS] [$v]$out"
done
