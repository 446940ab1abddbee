#!/bin/bash
# LightGlue linears with W on the accumulator-row side (float4 epilogues) against the round-5 orientation: tests first, then
# lg_bench and the B=64 step, alternating the two libraries on one box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/lg_swap_ab.txt
: > $O
timeout -k 10 900 python -m pytest tests/test_lightglue_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $O || { cat $O; exit 1; }
OLD=$GRAFT_REPO_ROOT/tools/experiments/ab_libs/libeinx_old.so
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export EINX_LIB=$OLD; else unset EINX_LIB; fi
    echo "== $lib $rep" >> $O
    python tools/lg_bench.py --only-linear --reps 10 2>/dev/null | tail -12 >> $O
    python bench.py --config sp_lg --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('sp_lg', d['value'], d['ms_per_step'], [ (s['stage'], s.get('ms'), s.get('frac')) for s in d.get('roofline_stages', [])] if isinstance(d.get('roofline_stages'), list) else d.get('roofline_stages'))
" >> $O
  done
done
unset EINX_LIB
cat $O
