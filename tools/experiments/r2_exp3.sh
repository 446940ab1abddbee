#!/bin/bash
set -o pipefail
O=gpurun_out/r2e3; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2e3/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [ (e['config'], e['pairs_per_step'], e['value'], e['ms_per_step']) for e in d.get('extra_configs',[])])
PY
