#!/bin/bash
# the stream picks of several launched one-rank runs (EINX_DEBUG_STREAMS) beside the post-hoc overlap report
cd $GRAFT_REPO_ROOT
O=gpurun_out/pg_debug.txt
: > $O
F="--no-cpu-baseline --no-extras --no-scale-legs --steps 40 --warmup 5"
for rep in 1 2 3 4 5 6; do
  echo "== spawn $rep" >> $O
  EINX_DEBUG_STREAMS=1 python bench.py --gpus 1 --spawn $F 2>&1 | grep -e "einx streams" -e '^{' | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d.get('streams'))
    elif 'cost' in l: print(l.strip())
" >> $O
done
cat $O
