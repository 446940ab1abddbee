#!/bin/bash
# round 5: conv parity tests on the working tree's library, then the conv layer table of two builds on ONE box
#   tools/experiments/r5_ab_layers.sh ab_libs/libeinx_r4.so [more libs...]   (run through gpurun from the repo root)
set -o pipefail
O=gpurun_out/r5_ab; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "conv" > $O/pytest_conv.txt 2>&1 || { tail -30 $O/pytest_conv.txt; exit 1; }
tail -3 $O/pytest_conv.txt
for rep in 1 2; do
  for L in "$@" ""; do
    tag=$(basename "${L:-tree}" .so)
    EINX_LIB=$L timeout -k 10 300 python bench.py --layer-table > $O/layers_${tag}_$rep.txt 2>&1 || { tail -20 $O/layers_${tag}_$rep.txt; exit 1; }
    echo "== $tag (run $rep): $(tail -1 $O/layers_${tag}_$rep.txt)"
  done
done
paste <(awk '{print $1, $(NF-3)}' $O/layers_libeinx_r4_2.txt) <(awk '{print $(NF-3)}' $O/layers_tree_2.txt) | column -t
