#!/bin/bash
# round 5: A/B of -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA accumulators in VGPRs: no v_accvgpr_read/write shuttling) on one box
#   tools/build_variant.sh vf "-mllvm -amdgpu-mfma-vgpr-form=1"; cp .../libeinx_hip.so ab_libs/libeinx_cur.so; gpurun -- tools/experiments/r5_vgpr_form.sh
set -o pipefail
O=gpurun_out/r5_vf; mkdir -p $O
for rep in 1 2; do
  for tag in cur vf; do
    L=ab_libs/libeinx_$tag.so
    EINX_LIB=$L timeout -k 10 300 python bench.py --layer-table > $O/layers_${tag}_$rep.txt 2>&1 || { tail -20 $O/layers_${tag}_$rep.txt; exit 1; }
    echo "== $tag (run $rep): $(tail -1 $O/layers_${tag}_$rep.txt)"
    EINX_LIB=$L timeout -k 10 300 python tools/lg_bench.py --skip-linear 2>&1 | tail -2
  done
done
paste <(awk '{print $1, $(NF-3)}' $O/layers_cur_2.txt) <(awk '{print $(NF-3)}' $O/layers_vf_2.txt)
