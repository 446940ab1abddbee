#!/bin/bash
# round 5: cache-policy bits on the un-pooled conv epilogue stores (first layers are store-issue-bound): layer table per variant, 3 processes each
set -o pipefail
O=gpurun_out/r5_aux; mkdir -p $O
for rep in 1 2 3; do
  for tag in cur aux2 aux17 aux18; do
    EINX_ALLOW_TIMING_ONLY=1 EINX_LIB=ab_libs/libeinx_$tag.so timeout -k 10 300 python bench.py --layer-table > $O/layers_${tag}_$rep.txt 2>&1 || { tail -20 $O/layers_${tag}_$rep.txt; exit 1; }
    echo "$tag run $rep: $(grep -E 'event.bb0|image.bb0|event.bb2|event.bb4|event.det0|total' $O/layers_${tag}_$rep.txt | awk '{printf "%s %s  ", $1, $(NF-3)}')"
  done
done
