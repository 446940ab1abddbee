#!/bin/bash
# the event extractor's side stream at normal / high priority: is the two-stream step still bimodal?  (SiLK B=32: 85 / 95 ms)
cd $GRAFT_REPO_ROOT
O=gpurun_out/prio_ab.txt
: > $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" >> $O 2>/dev/null
for cfg in silk_mnn sp_mnn; do
F="--config $cfg --no-cpu-baseline --no-extras --no-scale-legs --steps 10 --warmup 3"
for rep in 1 2 3 4 5 6; do
for v in 0 -1; do
  EINX_SIDE_PRIORITY=$v python bench.py $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$cfg prio $v', d['value'], d['ms_per_step'])
" >> $O
done
done
done
cat $O
