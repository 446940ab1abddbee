#!/bin/bash
# round 3, experiment 2: (a) store schedules for the dense-descriptor tensor, (b) does grid quantisation (the tail round of
# equal-length workgroups) explain the conv layers below conv1b?  layer table at batch sizes that make the rounds (nearly) whole
cd $GRAFT_REPO_ROOT
set -e
tools/bin/store_pattern2 | tee gpurun_out/r3e2_store.txt
for b in 32 38 19 48; do
  python bench.py --layer-table --batch $b 2>/dev/null | sed "s/^/B=$b: /" | tee -a gpurun_out/r3e2_layers.txt | grep -E "image.bb|image.de|total"
done
