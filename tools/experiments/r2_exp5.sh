#!/bin/bash
F="--no-cpu-baseline --no-extras --steps 40"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])"; }
for i in 1 2; do
python bench.py $F 2>/dev/null | q "plain                     "
python bench.py --spawn $F 2>/dev/null | q "spawn RCCL                "
GPU_MAX_HW_QUEUES=8 python bench.py --spawn $F 2>/dev/null | q "spawn RCCL, 8 hw queues   "
EINX_SIDE_PRIO=-1 python bench.py --spawn $F 2>/dev/null | q "spawn RCCL, side prio high"
EINX_SIDE_PRIO=-1 python bench.py $F 2>/dev/null | q "plain, side prio high     "
GPU_MAX_HW_QUEUES=8 python bench.py $F 2>/dev/null | q "plain, 8 hw queues        "
EINX_OVERLAP=0 python bench.py $F 2>/dev/null | q "plain, no overlap         "
done
