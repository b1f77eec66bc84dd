#!/bin/bash
# interleaved A/B of environment settings at one batch size: BATCH=6400 tools/ab_batch.sh "VAR=1" "VAR=2 OTHER=3" ...
# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
run() { r=$(env $1 python bench.py --steps ${STEPS:-200} --warmup 20 --repeats 3 --batch ${BATCH:-6400} ${SHAPE:-} --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); echo "[$BATCH: $1] $r"; }
for round in 1 2; do for v in "$@"; do run "$v"; done; done
