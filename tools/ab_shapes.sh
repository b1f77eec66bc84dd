#!/bin/bash
# interleaved A/B of environment settings over several shapes: tools/ab_shapes.sh "VAR=1" "VAR=2" ...   (SHAPES overrides the list)
# the experiment switches are read by the experiments build only (tuning.h): make -C cunvsm_amd/csrc dbg
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/cunvsm_amd/libcunvsm_amd_dbg.so}
cd "$(dirname "$0")/.."
SHAPES=${SHAPES:-"--batch=6400 --batch=12800 --batch=25600 --config=lse_small --batch=51200"}
for round in 1 2; do for sh in $SHAPES; do for v in "$@"; do
  r=$(env $v python bench.py --steps ${STEPS:-100} --warmup 20 --repeats 3 $sh --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "[$sh] [$v] $r"
done; done; done
