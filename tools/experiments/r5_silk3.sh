#!/bin/bash
# round 5: three workgroups per CU for the un-pooled 8x32 tile (3 spilled dwords at 80 registers) on the SiLK family, one box
for L in ab_libs/libeinx_cur.so "" ab_libs/libeinx_cur.so ""; do
  EINX_LIB=$L python bench.py --config silk_mnn --no-cpu-baseline --no-extras --no-scale-legs --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${L:-tree (3 per CU)}', d['value'], d['ms_per_step'])"
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "silk or conv_block" 2>&1 | tail -1
