#!/bin/bash
# round 5: wide NMS passes enqueued per forward (radius 4: images still changing afterwards are finished on the device), one box
for n in 8 4 3 8 4; do
  echo "== EINX_NMS_PASSES=$n"
  EINX_NMS_PASSES=$n python tools/latency_graph.py 2>&1 | grep sp_mnn
  EINX_NMS_PASSES=$n python bench.py --no-cpu-baseline --no-extras --no-scale-legs --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   B=32 headline', d['value'], d['ms_per_step'])"
done
