#!/bin/bash
# round 3, experiment 16: what would an LDS-DMA operand path buy the GEMM tile engine?  timing-only ablations (wrong results):
# gabl1 no register -> LDS commit, gabl2 also one barrier per K-slab, gabl3 also no global loads
cd $GRAFT_REPO_ROOT
for v in cur gabl1 gabl4 gabl3 cur gabl1 gabl4 gabl3; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  echo -n "$v: "; EINX_LIB=$L python tools/lg_bench.py --only-linear --reps 10 2>/dev/null | awk '{printf "%s %s %s us %s TF | ", $3, $4, $(NF-3), $(NF-1)} END {print ""}'
done
