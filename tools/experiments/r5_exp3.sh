#!/bin/bash
# round 5: workgroup placement probe + start-stagger variants of the conv kernels (lock-step hypothesis), one box
set -o pipefail
O=gpurun_out/r5_exp3; mkdir -p $O
timeout -k 10 60 tools/bin/wg_placement > $O/placement.txt 2>&1 || { tail $O/placement.txt; exit 1; }
run() { EINX_LIB=$2 timeout -k 10 300 python bench.py --layer-table > $O/layers_$1.txt 2>&1 || { tail -20 $O/layers_$1.txt; exit 1; }; echo "== $1: $(tail -1 $O/layers_$1.txt)"; }
run tree ei-nexus_official_amd/libeinx_hip.so
for v in stg1 stg2 stg1m1 stg1m2; do run $v ab_libs/libeinx_$v.so; done
run tree2 ei-nexus_official_amd/libeinx_hip.so
cd $O && for f in layers_*.txt; do echo "$f $(grep -E '^(event|image)\.' $f | awk '{printf "%s ", $(NF-3)}')"; done
