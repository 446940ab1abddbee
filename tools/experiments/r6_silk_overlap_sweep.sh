#!/bin/bash
# SiLK+MNN: two streams against one over the batch size, four runs each (the two-stream step is bimodal run to run)
cd $GRAFT_REPO_ROOT
O=gpurun_out/silk_overlap_sweep.txt
: > $O
for B in 1 2 4 8 16; do
F="--config silk_mnn --batch $B --no-cpu-baseline --no-extras --no-scale-legs --steps 20 --warmup 3"
for rep in 1 2 3 4; do
for v in two one; do
  unset EINX_OVERLAP; [ $v = one ] && export EINX_OVERLAP=0
  python bench.py $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B=$B $v', d['ms_per_step'])
" >> $O
done
done
done
python - <<'PY'
import collections
d = collections.defaultdict(list)
for l in open("gpurun_out/silk_overlap_sweep.txt"):
    b, v, ms = l.split()
    d[(b, v)].append(float(ms))
for k in d:
    print(k, sorted(d[k]))
PY
