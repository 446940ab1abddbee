#!/bin/bash
# SiLK family after the one-stream rule for large launches: six standalone runs each of SiLK+MNN B=32, B=1 and SiLK+LightGlue B=32
cd $GRAFT_REPO_ROOT
O=gpurun_out/silk_ab.txt
: > $O
for cfg in "silk_mnn --batch 32" "silk_mnn --batch 1" "silk_lg --batch 32"; do
for rep in 1 2 3 4 5 6; do
  python bench.py --config $cfg --no-cpu-baseline --no-extras --no-scale-legs --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$cfg', d['value'], d['ms_per_step'])
" >> $O
done
done
cat $O
