#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in cur prev cur prev; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python tools/nms_bench.py 2>/dev/null | tail -1 | sed "s/^/$v: /"
  EINX_LIB=$L python tools/lg_bench.py --skip-linear --reps 3 2>/dev/null | tail -1 | sed "s/^/$v: /"
  EINX_LIB=$L python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v sp_mnn', d['value'])"
done
