python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
for lay in 0 2 3 4; do for i in 1 2; do NVSM_SORT_LAYOUT=$lay python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('layout $lay', d['ms_per_step'], 'frac', r['frac'], 'loss_ms', r['avg_launch_ms'], 'gather frac', d['roofline_gather']['frac'])"; done; done
