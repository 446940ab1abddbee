#!/bin/bash
F="--no-cpu-baseline --no-extras --steps 40"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d.get('rccl'))"; }
python bench.py $F 2>/dev/null | q "plain      "
python bench.py --spawn $F 2>/dev/null | q "spawn RCCL "
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $F 2>/dev/null | q "torchrun 1 "
