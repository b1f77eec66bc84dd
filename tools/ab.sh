#!/bin/bash
# interleaved A/B of environment settings on the default bench: tools/ab.sh "VAR=1" "VAR=2 OTHER=1" ...  (ROUNDS rounds, default 3)
# PROFILE_FLAG= (empty) keeps the loss kernel's two events in the timed region (the default bench); default --no-profile
# prints ms per step and the loss kernel's roofline fraction; extra bench flags through BENCH_FLAGS
# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
for round in $(seq 1 ${ROUNDS:-3}); do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-extra-legs ${PROFILE_FLAG---no-profile} $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms  frac', (d.get('roofline') or {}).get('frac'), ' M/s', round(d['value']/1e6,2))")
    echo "round $round  [$cfg]  $r"
  done
done
