#!/bin/bash
# un-overlapped times of selected kernel groups under environment settings (experiments build): tools/r05_seq.sh "<bench flags>" "groups regex" "ENV=.." ...
cd "$(dirname "$0")/.."
export CUNVSM_AMD_LIB=${CUNVSM_AMD_LIB:-$PWD/cunvsm_amd/libcunvsm_amd_dbg.so}
FLAGS=$1; PAT=$2; shift 2
for v in "$@"; do
  echo "== [$FLAGS] [$v]"
  env $v python bench.py --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-extra-legs --sequential --profile-all $FLAGS 2>/dev/null | python -c "
import json,sys,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ms/step', d['ms_per_step'], ' '.join('%s=%.1f' % (k, v['avg_ms']*1e3) for k,v in d['kernel_breakdown'].items() if 'avg_ms' in v and re.search(r'$PAT', k)))
"
done
