#!/bin/bash
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; }
for v in remap noremap remap noremap; do
  if [ $v = remap ]; then L=""; else L="ab_libs/libeinx_noremap.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | grep -E "total" | sed "s/^/$v: /"
  EINX_LIB=$L python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | q "$v sp_mnn"
  EINX_LIB=$L python bench.py --config sp_lg --no-cpu-baseline --no-extras --steps 6 2>/dev/null | q "$v sp_lg "
  EINX_LIB=$L python tools/lg_bench.py --skip-linear --reps 3 2>/dev/null | tail -2 | sed "s/^/$v: /"
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
