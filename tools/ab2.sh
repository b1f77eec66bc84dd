#!/bin/bash
# as tools/ab.sh with extra bench flags in $BENCH_FLAGS
# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --no-profile $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "round $round  [$cfg]  $r ms"
  done
done
