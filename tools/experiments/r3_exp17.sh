#!/bin/bash
# round 3, experiment 17: separate occupancy from staging cost in the GEMM tile engine.
# gabl3 (no loads, no LDS writes; 64 VGPRs -> 4 workgroups/CU) with the persistent grid capped at 2 / 3 / 4 workgroups per CU,
# gabl4 (loads kept, no LDS writes; 102 VGPRs -> 2 per CU), cur (117 VGPRs -> 2 per CU)
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 per_cu=$2: "; EINX_GEMM_PER_CU=$2 EINX_LIB=$3 python tools/lg_bench.py --only-linear --reps 10 2>/dev/null | awk '{printf "%s %s %s us %s TF | ", $3, $4, $(NF-3), $(NF-1)} END {print ""}'; }
for r in 1 2; do
run cur 2 ""
run cur 1 ""
run gabl4 2 ab_libs/libeinx_gabl4.so
run gabl3 2 ab_libs/libeinx_gabl3.so
run gabl3 3 ab_libs/libeinx_gabl3.so
run gabl3 4 ab_libs/libeinx_gabl3.so
done
