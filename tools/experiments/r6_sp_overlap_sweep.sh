#!/bin/bash
# SP family: two streams against one over the batch size, five runs each (is the two-stream step bimodal here too?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/sp_overlap_sweep.txt
: > $O
for cfgB in "sp_lg 64" "sp_mnn 64" "sp_mnn 32" "sp_mnn 128"; do
set -- $cfgB
F="--config $1 --batch $2 --no-cpu-baseline --no-extras --no-scale-legs --steps 12 --warmup 3"
for rep in 1 2 3 4 5; do
for v in two one; do
  unset EINX_OVERLAP; [ $v = one ] && export EINX_OVERLAP=0
  python bench.py $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1_B$2 $v', d['ms_per_step'])
" >> $O
done
done
done
python - <<'PY'
import collections
d = collections.defaultdict(list)
for l in open("gpurun_out/sp_overlap_sweep.txt"):
    b, v, ms = l.split()
    d[(b, v)].append(float(ms))
for k in d:
    print(k, d[k])
PY
