#!/bin/bash
# round 3, experiment 1: conflict-free LDS pitch of the 3x3 conv tiles vs the plain pitch (ab_libs/libeinx_plainpitch.so),
# parity first, then the layer table A/B on one box, then the PMC passes of both builds.
cd $GRAFT_REPO_ROOT
set -e
python -m pytest tests/test_gpu_parity.py tests/test_random_shapes_gpu.py -x -q -m gpu -k "conv or random or extract or e2e" > gpurun_out/r3e1_tests.log 2>&1 || { tail -30 gpurun_out/r3e1_tests.log; exit 1; }
tail -3 gpurun_out/r3e1_tests.log
for v in cur plainpitch cur plainpitch; do
  if [ $v = cur ]; then L=""; else L="ab_libs/libeinx_$v.so"; fi
  EINX_LIB=$L python bench.py --layer-table 2>/dev/null | sed "s/^/$v: /" | tee -a gpurun_out/r3e1_layers.txt | grep -E "event.bb|image.det0|image.desc1|total"
done
OUT=pmc_conv3 bash tools/experiments/r3_pmc_conv.sh
OUT=pmc_conv3_plain EINX_LIB=$GRAFT_REPO_ROOT/ab_libs/libeinx_plainpitch.so bash tools/experiments/r3_pmc_conv.sh
