#!/bin/bash
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for v in cur thin4 noremap cur; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --config sp_lg --no-cpu-baseline --no-extras 2>/dev/null | q "$v sp_lg steps20"
  EINX_LIB=$L python bench.py --config sp_lg --no-cpu-baseline --no-extras --steps 6 2>/dev/null | q "$v sp_lg steps6 "
done
EINX_LIB="" python bench.py --config sp_lg --no-cpu-baseline --layer-table 2>/dev/null | tail -26
