for pr in 0 1 2 0 1 2; do
NVSM_AUX3_PRIO=$pr python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-legs --host-batches --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $pr host', d['ms_per_step'])"
NVSM_AUX3_PRIO=$pr python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('prio $pr', d['ms_per_step'], 'frac', r['frac'], 'loss_ms', r['avg_launch_ms'], 'gather frac', d['roofline_gather']['frac'])"; done
