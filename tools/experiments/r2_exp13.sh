#!/bin/bash
# (built with tools/build_variant.sh fat6 -DEINX_FAT_WAVES=6 while conv.hip had that launch-bounds knob; the knob was removed after this run)
cd $GRAFT_REPO_ROOT
for v in cur ${VARIANT:-fat6} cur ${VARIANT:-fat6}; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | grep -E "image.bb1|event.bb1|image.bb2|image.det0|total" | sed "s/^/$v: /"
  EINX_LIB=$L python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v sp_mnn', d['value'], d['roofline']['achieved'])"
done
