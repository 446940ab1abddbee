#!/bin/bash
# round 3, experiment 10: where do conv1b's idle matrix-core cycles go?  timing-only ablations (results are wrong).
# NOTE: the EINX_CONV_ABL=1..7 switches these variants were built with were removed from conv.hip after the run (numbers in
# profiles/r03_notes.md 3); -DEINX_CONV_ABL_NOLOADS=1 (= abl5) and -DEINX_CONV_DEPTH=2 are still there.
# abl1 no global staging (barriers kept), abl2 also no barriers, abl3 also no epilogue, abl4 full main loop, no epilogue
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-cur abl1 abl2 abl3 abl4 cur abl1 abl2 abl3 abl4}; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | grep -E "image.bb1|image.bb2|image.bb3|image.bb5|image.det0" | awk -v v=$v '{printf "%s %s %s us %s TF | ", v, $1, $(NF-3), $(NF-1)} END {print ""}'
done
