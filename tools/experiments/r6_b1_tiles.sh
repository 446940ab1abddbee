#!/bin/bash
# round 6: which tile the small-grid dispatcher should pick for the single-pair layers (conv1b at B=1: 67 us for 48 us of matrix time)
O=gpurun_out/r6_b1_tiles; mkdir -p $O
for v in "base" "EINX_PICK_POOLED=0" "EINX_PICK_POOLED=1" "EINX_PICK_POOLED=2" "EINX_PICK_POOLED=3" "EINX_T16_MAX=16384" "EINX_T16_MAX=40000"; do
  n=$(echo $v | tr '=' '_')
  if [ "$v" = "base" ]; then python bench.py --layer-table --batch 1 > $O/$n.txt 2>&1; else env $v python bench.py --layer-table --batch 1 > $O/$n.txt 2>&1; fi
  echo "== $v"; grep "image\.\(bb\|det\|desc\)\|total" $O/$n.txt | awk '{printf "%s %s %s %s | ", $1, $6, $8, $9} END {print ""}'
done
