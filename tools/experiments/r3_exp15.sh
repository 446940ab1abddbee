#!/bin/bash
# round 3, experiment 15: reference-complete dict, dense kernels scheduled beside the other extractor's convolutions / the matcher
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "dense or e2e or golden or install or properties" 2>&1 | tail -2
for v in 1 0 1 0; do  # 0 = shipped
  echo -n "EINX_DENSE_SCHEDULE=$v: "; EINX_DENSE_SCHEDULE=$v python bench.py --dense --log-assignment --no-cpu-baseline --no-extras --no-scale-legs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
