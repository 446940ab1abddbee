#!/bin/bash
# round 5: all GPU tests, the conv layer table and the default bench line of the working tree (one box)
set -o pipefail
O=gpurun_out/r5_full; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -40 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py --layer-table > $O/layers.txt 2>&1 || { tail -20 $O/layers.txt; exit 1; }
tail -1 $O/layers.txt
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_full/bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['launch_ms'])
for s in d.get('roofline_stages',[]): print(' ', s['stage'][:60], s.get('ms'), s.get('frac'))
for e in d.get('extra_configs',[]): print(' ', e['config'], e['workload'][:40], e['value'], e['ms_per_step'])
PY
