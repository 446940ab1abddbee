#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2e7; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in default thin8 thin4; do
  if [ $v = default ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | grep -E "bb0|total" | sed "s/^/$v: /"
done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['roofline'])[:700])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ko -o p -- python3 $GRAFT_REPO_ROOT/bench.py --kernel-only > $O/ko.log 2>&1
grep -h "8, true, true" $(find $O/prof_ko -name "*kernel_stats.csv") | cut -c1-200
tail -2 $O/ko.log | cut -c1-900
