#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_random_shapes_gpu.py tests/test_gpu_parity.py -x -q -k "conv or extractors_small or e2e" 2>&1 | tail -3
for e in 32 0; do
  echo "== EINX_CONV_EXP=$e (32 = small-grid mode off)"
  EINX_CONV_EXP=$e python tools/latency_b1.py 1 2>&1 | tail -1
  EINX_CONV_EXP=$e python tools/profile_b.py 1 2>/dev/null | grep -E "total conv|bb1|bb5|bb6|det0" | head -12
  EINX_CONV_EXP=$e python tools/latency_b1.py 2 2>&1 | tail -1
  EINX_CONV_EXP=$e python tools/latency_b1.py 4 2>&1 | tail -1
done
python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=32', d['value'])"
