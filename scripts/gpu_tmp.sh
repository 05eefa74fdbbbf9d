export TMPDIR=/tmp
python scripts/gpu_box_probe.py c2 c5 c4 c1 2>&1 | grep BOX
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
timeout 400 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench default:', round(d['value']), 'norm', round(d['value_normalised']), 'ms', round(d['ms_per_step'],3), 'unsolved', d['unsolved'], 'stale', d['roofline'].get('traffic_stale'), 'gpu/cpu', d.get('gpu_over_cpu'))"
