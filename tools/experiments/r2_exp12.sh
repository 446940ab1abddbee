#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "conv_block or extractors_small or e2e_full" 2>&1 | tail -3
for v in cur fw6; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  for e in 16 0 512 2048; do   # 16: generic kernel; tpw = 4 (default), 2, 8
    EINX_LIB=$L EINX_CONV_EXP=$e python bench.py --layer-table 2>/dev/null | grep -E "bb0" | sed "s/^/$v exp=$e: /"
  done
done
for e in 16 0; do
  EINX_CONV_EXP=$e python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp=$e sp_mnn', d['value'])"
  EINX_CONV_EXP=$e python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp=$e sp_mnn', d['value'])"
done
