#!/bin/bash
# interleaved A/B of environment settings with the loss kernel's in-step time next to the step time: tools/ab_roof.sh "VAR=1" "VAR=2" ...
# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
for round in 1 2; do for v in "$@"; do
  env $v python bench.py --steps ${STEPS:-60} --warmup 20 --repeats 3 ${SHAPE:-} --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v]', d['ms_per_step'], 'loss', r['avg_launch_ms'], 'frac', r['frac'])"
done; done
