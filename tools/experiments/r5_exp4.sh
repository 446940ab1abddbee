#!/bin/bash
# round 5: timing-only ablations of the new conv kernel (what the matrix pipe's idle 12 % is made of), one box
set -o pipefail
O=gpurun_out/r5_exp4; mkdir -p $O
run() { EINX_ALLOW_TIMING_ONLY=1 EINX_LIB=$2 timeout -k 10 300 python bench.py --layer-table > $O/layers_$1.txt 2>&1 || { tail -20 $O/layers_$1.txt; exit 1; }; echo "== $1: $(tail -1 $O/layers_$1.txt)"; }
run tree ei-nexus_official_amd/libeinx_hip.so
for v in abl1 abl2 abl4 abl7; do run $v ab_libs/libeinx_$v.so; done
run wps6 ab_libs/libeinx_wps6.so
cd $O && for f in layers_*.txt; do echo "$f $(grep -E '^(event|image)\.' $f | awk '{printf "%s ", $(NF-3)}')"; done
