#!/bin/bash
# un-overlapped per-kernel-group times (sequential calls, events around every group): tools/seq_breakdown.sh [bench flags]
cd "$(dirname "$0")/.."
python bench.py --steps 50 --warmup 10 --repeats 1 --no-cpu-baseline --no-extra-legs --sequential --profile-all "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'])
for k,v in d['kernel_breakdown'].items():
    if 'avg_ms' in v: print('  %-22s %7.1f us' % (k, v['avg_ms']*1e3), v.get('TFLOPs',''), v.get('algorithmic_GBps',''))
"
