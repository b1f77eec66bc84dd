#!/bin/bash
# round-5 baseline probe (GPU box): free-running timelines and un-overlapped kernel times of the per-rank and LSE shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_probe
O=gpurun_out/r05_probe
q() { python bench.py --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timing']['ms_per_step_all'])"; }
for sh in "--batch=6400" "--config=lse_small" "--batch=51200" "--batch=12800"; do echo "[$sh] $(q $sh)"; done > $O/base.txt 2>&1
tools/seq_breakdown.sh --batch 6400 > $O/seq_6400.txt 2>&1
tools/seq_breakdown.sh --config lse_small > $O/seq_lse.txt 2>&1
tools/timeline_free.sh r05_6400 --batch 6400 > /dev/null 2>&1; cp gpurun_out/timeline_r05_6400.txt $O/
tools/timeline_free.sh r05_lse --config lse_small > /dev/null 2>&1; cp gpurun_out/timeline_r05_lse.txt $O/
cat $O/base.txt
