#!/bin/bash
# A/B on ONE box: does an initialised RCCL process group slow the step loop?
O=gpurun_out/r2e4; mkdir -p $O
F="--no-cpu-baseline --no-extras --steps 40"
for i in 1 2; do
python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain        ', d['value'])"
EINX_BENCH_NO_PG=1 python bench.py --spawn $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spawn, no PG ', d['value'])"
python bench.py --spawn $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spawn, RCCL  ', d['value'], d['rccl'])"
TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 python bench.py --spawn $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spawn, RCCL, no watchdog monitoring', d['value'])"
done
