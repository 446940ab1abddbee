#!/bin/bash
# round 3, experiment 9: lockstep hypothesis -- a one-off start delay for the second resident round of conv workgroups
cd $GRAFT_REPO_ROOT
for v in cur stag2 stag5 stag10 cur stag2 stag5 stag10; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | grep -E "image.bb1|image.bb2|image.bb3|image.bb5|image.det0|total" | awk -v v=$v '{printf "%s %s %s us %s TF | ", v, $1, $(NF-3), $(NF-1)} END {print ""}'
done
